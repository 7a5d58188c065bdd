// host_overlap.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): the overlap core (OverlapRun: seeds, batch plan, expand, sort, groups, chain, count) and its entry points: two-set, inverse, all-vs-all, chains, PAF statistics.
// ------------------------------------------------------------------------------------------
// overlap core
// ------------------------------------------------------------------------------------------
enum { MODE_TWOSET = 0, MODE_INVERSE = 1, MODE_AVA = 2 };

// The streamed set's sketch, kept across the parts of a partitioned index (every part sees the same queries: sketched once, not
// once per part -- 8 x 12.7 ms at full-size C5)
struct SketchCache { std::unique_ptr<Scratch> sc; SketchOut so; std::vector<u32> h_mzoff; bool valid = false; };

struct OverlapJob {
    int mode;
    int dual;                       // 1: NO_DUAL cleared, 0: set
    lrge_hip_params prm;
    // outputs (host)
    u32 *counts = nullptr;          // size: nq (twoset) or n_indexed (inverse / ava)
    u32 *has_map = nullptr;
    lrge_hip_chain *chains = nullptr; u64 chain_cap = 0; u64 *n_chains = nullptr;
    // anchors of one query instead of chaining
    bool dump_anchors = false; u32 dump_query = 0; u64 *ax = nullptr, *ay = nullptr; u64 acap = 0; u64 *an = nullptr;
    // per-query PAF statistics instead of chaining
    bool paf_stats = false; i32 *rep_len = nullptr; u64 *sum_span = nullptr; u32 *n_kept = nullptr;
    // one part of a partitioned index (the entry points loop over the parts)
    u32 rid_base = 0;                                   // first read of the part in the whole indexed set
    const lrge_hip_seqset *indexed_top = nullptr;       // all-vs-all: the whole indexed set (counts are keyed by it)
    u32 *d_hc_acc = nullptr; bool hc_last = true;       // paf_stats: occurrence counts accumulated over the parts (device)
    const u32 *d_hc_global = nullptr;                   // chain records: those counts, complete (a seed's rank among the KEPT seeds
                                                        // of its query -- n_seeds / dv -- counts seeds kept in ANY part)
    SketchCache *qcache = nullptr;                      // the streamed set's sketch, shared by the parts' runs
};


// Split of the size-sorted group list between the two chain kernels, from the size census of k_group_count (hn / ha:
// groups and anchors per class of GSZ_W anchors).  Groups above T anchors -> k_chain_hw (~0.55 us per anchor of latency,
// ~93 VALU instructions per anchor), the rest -> k_chain_lpg (~4.1 us per anchor of the LONGEST group of a wavefront,
// ~21 VALU per anchor).  Both run side by side; the stage takes about
//   max(T * t_lpg, n_longest * t_hw, VALU work / issue rate of the chip)
// and T (a multiple of GSZ_W) minimises that estimate -- measured constants of this kernel pair on MI355X.
// `fixed` != LPG_MAX_AUTO pins T (LRGE_HIP_LPG_MAX / LRGE_HIP_CHAIN=hw|lpg).
struct ChainSplit { u32 T, n_big; unsigned long long a_big; int top; };
static ChainSplit choose_chain_split(const u32 *hn, const unsigned long long *ha, unsigned long long a_chained, u32 fixed, int n_cu) {
    ChainSplit r; r.T = fixed; r.n_big = 0; r.a_big = 0; r.top = -1;
    for (int b = 0; b < GSZ_BINS; ++b) if (hn[b]) r.top = b;
    if (fixed == LPG_MAX_AUTO) {
        // measured constants of this kernel pair on MI355X.  `rate` is the wave64 VALU instruction rate the chip sustains for
        // k_chain_lpg at its residency (1.25 wavefronts per SIMD, bounded by LDS) -- 422 G/s measured at C4; a shape with four
        // wavefronts per workgroup and twice the residency was measured too (round 2): every step took 1.4x as long and the
        // stage was slower or equal on C2, C4 and C5/10 alike, because the stage is bound by T * t_lpg, not by throughput
        const double t_lpg = 4.1e-6, t_hw = 0.55e-6, c_lpg = 21.0, c_hw = 93.0;
        const double rate = 0.8 * (double)n_cu * 4 * 2.1e9 / 4.0;
        double best = 1e30, a_le = 0;    // a_le: anchors in classes <= b
        r.T = 0;
        for (int b = -1; b < GSZ_BINS - 1; ++b) {      // T = (b + 1) * GSZ_W: classes 0..b go to k_chain_lpg
            if (b >= 0) a_le += (double)ha[b];
            const double a_hw = (double)a_chained - a_le;
            const double crit_lpg = b >= 0 ? (double)std::min<int>(b + 1, r.top + 1) * GSZ_W * t_lpg : 0.0;
            const double crit_hw = a_hw > 0 ? (double)(r.top + 1) * GSZ_W * t_hw : 0.0;
            const double est = std::max(std::max(crit_lpg, crit_hw), (a_le * c_lpg + a_hw * c_hw) / rate);
            if (est < best - 1e-9) { best = est; r.T = (u32)(b + 1) * GSZ_W; }
            if (b >= r.top) break;
        }
    }
    // groups above T: whole classes (class b = (b*W, (b+1)*W]); a pinned T that is no class edge counts by class floor --
    // any split point of the sorted list is valid, only the balance depends on it
    for (int b = 0; b < GSZ_BINS; ++b)
        if ((u64)b * GSZ_W >= (u64)r.T) { r.n_big += hn[b]; r.a_big += ha[b]; }
    return r;
}

// One overlap call = one OverlapRun: the state every stage shares lives here, the stages are its methods
// (prepare -> seeds -> plan -> batch x N -> finish); a stage returns RUN_DONE when the call is complete early
// (empty sets, statistics-only or anchor-dump runs).
enum { RUN_DONE = 1 };

struct OverlapRun {
    lrge_hip_ctx *ctx; const lrge_hip_index *ix; const lrge_hip_seqset *Q; OverlapJob &job;
    Scratch sc;
    // outputs on the device
    u32 n_out = 0; u32 *d_qmap = nullptr, *d_counts = nullptr, *d_hasmap = nullptr;
    unsigned long long *d_nchains = nullptr; lrge_hip_chain *d_chains = nullptr;
    bool need_rank = true;      // seed ranks (krank) are wanted by this run's anchors
    // seeds: query minimizers, their index lookups, per-query anchor totals
    SketchOut so; std::vector<u32> h_mzoff, h_qtot; u64 Mq = 0; SeedParams sp;
    std::unique_ptr<Scratch> presk_sc;   // memory of a consumed presketch (released with the run)
    u64 *hs = nullptr;                   // where every seed's list lives: start in pos[], or HT_INLINE | y (k_index.h)
    u32 *hc = nullptr, *hn = nullptr, *hv = nullptr, *krank = nullptr, *aoff_all = nullptr;
    // batch plan
    u64 batch_cap = 0; KeyLayout kl; u32 max_bits_q = 0, min_n = 0; ChainParams cp;
    std::vector<SegTile> h_tiles;   // per batch; lives until the batch's next host sync (the async H2D copy reads it)
    std::vector<SegDesc> h_local[3];
    u32 n_local_items = 0;         // anchors of the batch sorted by k_seg_sort_local
    std::vector<u32> h_qkept, h_qlist;   // dead-pair filter (k_expand_q): anchors every query of the batch kept; the queries by size class

    OverlapRun(lrge_hip_ctx *c, const lrge_hip_index *i, const lrge_hip_seqset *q, OverlapJob &j) : ctx(c), ix(i), Q(q), job(j), sc(c) {}
    int prepare();                              // output buffers, shard map, empty-set shortcut
    int seeds();                                // K1 sketch, K3 lookup, K4a query-occurrence filter, hit counts
    int plan();                                 // batch size, key layout, chaining parameters
    int batch(u32 q0, u32 q1, u64 A);           // K4 expand, sort, K5 groups, K6 chain, K7 count for queries [q0, q1)
    int finish();                               // results to the host
    void plan_anchor_sort(u32 q0, u32 q1, bool packed, const u32 *kept);   // which queries sort inside LDS, tiles for the rest
    int dump_sorted_anchors(const u64 *skey, const u64 *sval, u64 A);   // lrge_hip_anchors_dump: one query's anchors, mm2 encoding
};

int OverlapRun::prepare() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    const lrge_hip_seqset *I = (job.mode == MODE_AVA && job.indexed_top) ? job.indexed_top : T;   // what the counts are keyed by
    n_out = job.mode == MODE_TWOSET ? nq : (job.mode == MODE_AVA ? I->n : nt);
    if (job.mode == MODE_AVA && Q != I) {
        // a shard of the reads as queries: counts stay keyed by indexed read, so every query needs the index of the
        // read with the same name (= the same rank) in the indexed set
        const u32 ni = I->n;
        std::vector<std::pair<u32, u32>> byrank(ni);
        for (u32 i = 0; i < ni; ++i) byrank[i] = {I->h_rank[i], i};
        std::sort(byrank.begin(), byrank.end());
        std::vector<u32> qm(nq);
        for (u32 q = 0; q < nq; ++q) {
            auto it = std::lower_bound(byrank.begin(), byrank.end(), std::make_pair(Q->h_rank[q], 0u));
            if (it == byrank.end() || it->first != Q->h_rank[q]) { LRGE_SET_ERR(ctx, "all-vs-all shard: read %u is not in the indexed set", q); return LRGE_ERR_INVALID; }
            qm[q] = it->second;
        }
        d_qmap = sc.get<u32>((size_t)nq + 1);
        if (!d_qmap) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(d_qmap, qm.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // qm is a local
    }
    d_counts = sc.get<u32>((size_t)n_out + 1); d_hasmap = sc.get<u32>((size_t)nq + 1);
    if (!d_counts || !d_hasmap) return LRGE_ERR_DEVICE;
    HIPCHK(ctx, hipMemsetAsync(d_counts, 0, ((size_t)n_out + 1) * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_hasmap, 0, ((size_t)nq + 1) * 4, ctx->stream));
    if (job.n_chains) {
        d_nchains = (unsigned long long *)sc.get<u64>(1);
        d_chains = sc.get<lrge_hip_chain>(job.chain_cap ? job.chain_cap : 1);
        if (!d_nchains || !d_chains) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemsetAsync(d_nchains, 0, 8, ctx->stream));
    }
    ctx->counters[LRGE_C_QUERY_BASES] = Q->total_bases;
    if (nq == 0 || nt == 0) {
        if (job.counts) memset(job.counts, 0, (size_t)n_out * 4);
        if (job.has_map) memset(job.has_map, 0, (size_t)nq * 4);
        if (job.n_chains) *job.n_chains = 0;
        if (job.an) *job.an = 0;
        if (job.paf_stats) { memset(job.rep_len, 0, (size_t)nq * 4); memset(job.sum_span, 0, (size_t)nq * 8); memset(job.n_kept, 0, (size_t)nq * 4); }
        return RUN_DONE;
    }
    return LRGE_OK;

}

int OverlapRun::seeds() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    // ---- 1. sketch the queries ----
    int rc = LRGE_OK;
    if (Q->presk && Q->presk->preset == ix->preset_id) {
        // sketched ahead on the side stream (lrge_hip_seqset_presketch): wait for it on the device, fetch the count and
        // the per-read offsets in the one round trip the in-line sketch pays too, and keep its memory until the call ends
        PreSketch *p = Q->presk;
        const_cast<lrge_hip_seqset *>(Q)->presk = nullptr;
        presk_sc.reset(p->sc);
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, p->ev_done, 0));
        u32 total = 0;
        h_mzoff.resize((size_t)Q->n + 1);
        HIPCHK(ctx, ctx->d2h(&total, p->d_total, 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h(h_mzoff.data(), p->mz_off, ((size_t)Q->n + 1) * 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        so.x = p->x; so.y = p->y; so.mz_off = p->mz_off; so.n = total;
        ctx->timers.push_back(TimerRec{LRGE_T_SKETCH, p->ev_start, p->ev_done});   // both have completed; resolved with the call's timers
        delete p;
        if (job.qcache) {     // (the other parts of a partitioned index reuse it)
            job.qcache->sc = std::move(presk_sc); job.qcache->so = so; job.qcache->h_mzoff = h_mzoff; job.qcache->valid = true;
        }
    } else if (job.qcache && job.qcache->valid) {
        so = job.qcache->so; h_mzoff = job.qcache->h_mzoff;
    } else {
        if (job.qcache && !job.qcache->sc) job.qcache->sc.reset(new Scratch(ctx));
        rc = sketch_device(ctx, job.qcache ? *job.qcache->sc : sc, Q, ix->preset_id, false, &so, 0, 0, &h_mzoff);
        if (rc) return rc;
        if (job.qcache) { job.qcache->so = so; job.qcache->h_mzoff = h_mzoff; job.qcache->valid = true; }
    }
    Mq = so.n;
    ctx->counters[LRGE_C_QUERY_MINIMIZERS] = Mq;
    if (Mq >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query set limited to < 2^32 minimizers"); return LRGE_ERR_TOO_MANY; }

    // ---- 2. lookup ----
    sp.ht = ix->d_ht; sp.ht_cap = ix->ht_cap; sp.ht_fix = ix->ht_fix; sp.pos = ix->d_pos; sp.pk_pos1 = ix->pk_pos1; sp.pk_ybits = ix->pk_ybits;
    sp.t_len = T->d_len; sp.t_rank = T->d_rank; sp.q_len = Q->d_len; sp.q_rank = Q->d_rank;
    sp.mid_occ = ix->mid_occ;
    sp.check_names = (Q->has_rank && T->has_rank) ? 1 : 0;   // qname == NULL in minimap2 skips skip_seed entirely
    if (sp.check_names && job.dual) {
        // with --dual=yes skip_seed only ever fires for a query that IS one of the indexed reads (same name, same
        // length, same position).  Ranks are positions in the sorted union of names, so if no rank occurs in both
        // sets (the two-set strategies) no hit can be skipped and the per-hit name checks are dropped altogether.
        const bool shared = ranks_intersect(Q->h_rank, T->h_rank);
        if (!shared) sp.check_names = 0;
    }
    sp.no_dual = job.dual ? 0 : 1;
    // without name checks every kept hit survives skip_seed: hv IS hn, and k_lookup fills it (k_seed_counts only runs again
    // if the exact query occurrence filter had to change hc)
    const bool counts_in_lookup = !sp.check_names && !ctx->opt("COUNTS_AFTER_LOOKUP");   // (option: the separate pass, for A/B runs)
    hs = sc.get<u64>(Mq + 1); hc = sc.get<u32>(Mq + 1); hn = sc.get<u32>(Mq + 1); hv = counts_in_lookup ? hn : sc.get<u32>(Mq + 1); krank = sc.get<u32>(Mq + 1);
    u32 *d_qtot = sc.get<u32>((size_t)nq + 1);
    aoff_all = sc.get<u32>(Mq + 1);
    if (!hs || !hc || !hn || !hv || !krank || !d_qtot || !aoff_all) return LRGE_ERR_DEVICE;
    h_qtot.assign((size_t)nq + 1, 0);
    if (Mq) {
        StageTimer t(ctx, LRGE_T_LOOKUP), tk(ctx, LRGE_T_K_LOOKUP);
        hipLaunchKernelGGL(k_lookup, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.x, Mq, sp, hs, hc, counts_in_lookup ? hn : (u32 *)nullptr);
        KCHK(ctx);
        tk.stop(); t.stop();
        ctx->counters[LRGE_C_LOOKUP_LAUNCHES] += 1;
    }

    // ---- 3. query occurrence filter (mm_seed_mz_flt) ----
    // minimap2 applies it before the lookup; the result is the same afterwards, restricted to the
    // minimizers present in the index: every occurrence of a value x in one query gets the same lookup
    // result, so the per-query multiplicity of x is fully visible inside that subset, and absent values
    // contribute nothing whether removed or not.  A removed minimizer is marked absent (hc = 0).
    // The exact filter (two radix sorts + a run-length mark) as a callable: it only runs when the conservative
    // pre-check k_qocc_check cannot rule it out, or when LRGE_HIP_QOCC_EXACT forces it (tests).
    const u32 *d_qsel = nullptr;      // per-query verdicts of the pre-check (null: the exact pass takes every query)
    auto run_exact_qocc = [&]() -> int {
        StageTimer t(ctx, LRGE_T_QFILTER);
        ALLOC_OR_FAIL(flag, sc, u32, Mq); ALLOC_OR_FAIL(fpos, sc, u32, Mq); ALLOC_OR_FAIL(d_ns, sc, u32, 1);
        hipLaunchKernelGGL(k_flag_present_sel, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hc, so.y, d_qsel, Mq, flag);
        KCHK(ctx);
        rc = scan_exclusive_u32(ctx, sc, flag, fpos, Mq, d_ns);
        if (rc) return rc;
        u32 Ms = 0;
        HIPCHK(ctx, ctx->d2h(&Ms, d_ns, 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        if (Ms > (u32)ix->mid_occ) {
            ALLOC_OR_FAIL(ka, sc, u64, Ms); ALLOC_OR_FAIL(va, sc, u64, Ms);
            ALLOC_OR_FAIL(kb, sc, u64, Ms); ALLOC_OR_FAIL(vb, sc, u64, Ms);
            hipLaunchKernelGGL(k_qocc_keys, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.x, so.y, flag, fpos, Mq, ka, va);
            KCHK(ctx);
            u64 *rk, *rv;
            rc = radix_sort_pairs(ctx, sc, ka, va, kb, vb, Ms, 0, 2 * P.k + 8, &rk, &rv);   // by x
            if (rc) return rc;
            u64 *ok = (rk == ka) ? kb : ka, *ov = (rv == va) ? vb : va;
            u64 *rk2, *rv2;
            // then (stable) by query id held in bits [32, 32+bits) of the value: swap roles
            rc = radix_sort_pairs(ctx, sc, rv, rk, ov, ok, Ms, 32, (int)ceil_log2_u64((u64)nq + 1), &rk2, &rv2);
            if (rc) return rc;
            hipLaunchKernelGGL(k_qocc_mark, dim3((u32)div_up(Ms, 256)), dim3(256), 0, ctx->stream, rv2 /* x */, rk2 /* (q,idx) */,
                               (u64)Ms, so.mz_off, ix->mid_occ, P.q_occ_frac, hc);
            KCHK(ctx);
            sc.drop(ka); sc.drop(va); sc.drop(kb); sc.drop(vb);
        }
        sc.drop(flag); sc.drop(fpos); sc.drop(d_ns);
        t.stop();
        return LRGE_OK;
    };
    bool qocc_possible = false;
    bool hc_changed = false;       // by run_exact_qocc: k_lookup's own kept counts are stale then
    if (Mq > 0 && P.q_occ_frac > 0.0f && ix->mid_occ > 0)   // only queries with more minimizers than mid_occ can be affected
        for (u32 q = 0; q < nq && !qocc_possible; ++q) qocc_possible = (i64)(h_mzoff[q + 1] - h_mzoff[q]) > (i64)ix->mid_occ;
    u32 *d_qf = nullptr; u32 qf = 0; bool qf_on_side = false;
    if (qocc_possible) {
        if (ctx->opt("QOCC_EXACT")) { rc = run_exact_qocc(); if (rc) return rc; hc_changed = true; }
        else {
            // cheap conservative check, on the side stream beside the hit counting below (both only read the lookup
            // results); its verdict travels to the host with the next sync (no extra round trip)
            d_qf = sc.get<u32>((size_t)nq + 1);     // [0] any query, [1 + q] query q
            if (!d_qf) return LRGE_ERR_DEVICE;
            d_qsel = d_qf;
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            StageTimer t(ctx, LRGE_T_QFILTER, ctx->stream2);
            HIPCHK(ctx, hipMemsetAsync(d_qf, 0, ((size_t)nq + 1) * 4, ctx->stream2));
            hipLaunchKernelGGL(k_qocc_check, dim3(nq), dim3(256), 0, ctx->stream2, so.x, hc, so.mz_off, nq, ix->mid_occ, d_qf);
            KCHK(ctx);
            t.stop();
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            qf_on_side = true;
        }
    }
    if (job.paf_stats) {   // per-query seed statistics only (rl, avg_k ingredients)
        if (d_qf) {
            if (qf_on_side) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); qf_on_side = false; }
            HIPCHK(ctx, ctx->d2h(&qf, d_qf, 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            if (qf) { rc = run_exact_qocc(); if (rc) return rc; }
        }
        const u32 *hc_stats = hc;
        if (job.d_hc_acc) {      // one part of a partitioned index: the statistics need the counts over all parts
            if (Mq) { hipLaunchKernelGGL(k_hc_accumulate, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hc, Mq, (u32)ix->mid_occ, job.d_hc_acc); KCHK(ctx); }
            if (!job.hc_last) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return RUN_DONE; }
            hc_stats = job.d_hc_acc;
        }
        ALLOC_OR_FAIL(d_rl, sc, i32, (size_t)nq); ALLOC_OR_FAIL(d_ss, sc, u64, (size_t)nq); ALLOC_OR_FAIL(d_nk, sc, u32, (size_t)nq);
        hipLaunchKernelGGL(k_query_paf_stats, dim3((u32)div_up(nq, 64)), dim3(64), 0, ctx->stream, so.x, so.y, hc_stats, so.mz_off, nq, ix->mid_occ,
                           d_rl, d_ss, d_nk);
        KCHK(ctx);
        HIPCHK(ctx, hipMemcpyAsync(job.rep_len, d_rl, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(job.sum_span, d_ss, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(job.n_kept, d_nk, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return RUN_DONE;
    }
    auto run_counts = [&]() -> int {
        StageTimer t(ctx, LRGE_T_LOOKUP);
        if (Mq) {
            if (!counts_in_lookup || hc_changed) {
                hipLaunchKernelGGL(k_seed_counts, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.y, Mq, sp, hs, hc, hn, hv);
                KCHK(ctx);
            }
            // rank of every kept seed inside its query (= its index in minimap2's mini_pos[]): only chain records carry it
            // (mm_est_err's dv); a count-only run packs its anchors without it (OverlapRun::batch) and skips the flag + scan
            const u32 bits_rpos_ = std::max<u32>(1, ceil_log2_u64((u64)T->max_len + 1)), bits_rid_ = std::max<u32>(1, ceil_log2_u64((u64)T->n));
            const u32 bits_qy_ = std::max<u32>(1, ceil_log2_u64((u64)Q->max_len + 1));
            need_rank = d_chains || job.dump_anchors || bits_rpos_ + 1 + bits_rid_ + bits_qy_ + 9 > 64 || ctx->opt_u64("NO_PACKED", 0);
            if (need_rank) {
                ALLOC_OR_FAIL(kflag, sc, u32, Mq);
                if (job.d_hc_global) hipLaunchKernelGGL(k_flag_kept, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, job.d_hc_global, Mq, (u32)ix->mid_occ, kflag);
                else hipLaunchKernelGGL(k_flag_nonzero, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hn, Mq, kflag);
                KCHK(ctx);
                rc = scan_exclusive_u32(ctx, sc, kflag, krank, Mq, krank + Mq);
                if (rc) return rc;
                sc.drop(kflag);
            }
        } else {
            HIPCHK(ctx, hipMemsetAsync(krank, 0, 4, ctx->stream));
        }
        // ONE scan of the surviving-hit counts over all query minimizers: the per-query totals are differences of it, and every
        // batch's k_expand reads its output offsets from it (relative to the batch's first minimizer; all modulo 2^32, so a job
        // with more than 2^32 anchors is fine as long as a batch -- at most 2^30 -- and a query stay below)
        if (Mq) {
            rc = scan_exclusive_u32(ctx, sc, hv, aoff_all, Mq, aoff_all + Mq);
            if (rc) return rc;
        } else HIPCHK(ctx, hipMemsetAsync(aoff_all, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_query_totals_from_scan, dim3((u32)div_up(nq, 256)), dim3(256), 0, ctx->stream, aoff_all, so.mz_off, nq, d_qtot);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(h_qtot.data(), d_qtot, (size_t)nq * 4, ctx->stream));
        if (d_qf) {
            if (qf_on_side) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); qf_on_side = false; }
            HIPCHK(ctx, ctx->d2h(&qf, d_qf, 4, ctx->stream));
        }
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        t.stop();
        return LRGE_OK;
    };
    rc = run_counts();
    if (rc) return rc;
    if (d_qf && qf) {   // the pre-check could not rule the filter out: apply it, then count again
        d_qf = nullptr;
        rc = run_exact_qocc(); if (rc) return rc;
        hc_changed = true;
        rc = run_counts(); if (rc) return rc;
    }
    return LRGE_OK;
}

int OverlapRun::plan() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    // ---- 4. batches ----
    // Anchors per batch.  Every batch pays the latency of its longest chain group once (the chain kernels are bound by it:
    // T steps of ~4 us whatever the batch holds), so batches are as large as memory and the 32-bit anchor positions allow: 2^31
    // anchors (full-size C5: one batch per index part / per view instead of two -- 15 -> 8 batches, chain 273 -> 184 ms,
    // inverse step 1.135 -> 1.041 s, counts identical), ~40 B of scratch per anchor at the peak (four 8-byte arrays through the
    // sort; 16 + group starts + records + marks behind it), budgeted as 48 B out of 4/5 of the free HBM.
    batch_cap = 1ULL << 31;
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
            const u64 by_mem = ((u64)mfree + ctx->pool.idle()) / 5 * 4 / 48;     // the pool's idle blocks are reusable too (not the ones in use: a resident index)
            if (by_mem < batch_cap) batch_cap = by_mem;
        }
        if (batch_cap < (1ULL << 20)) batch_cap = 1ULL << 20;
    }
    batch_cap = ctx->opt_u64("BATCH_ANCHORS", batch_cap);
    kl.bits_rpos = std::max<u32>(1, ceil_log2_u64((u64)T->max_len + 1));
    kl.bits_rid = std::max<u32>(1, ceil_log2_u64((u64)nt));
    max_bits_q = 63 - (kl.bits_rpos + 1 + kl.bits_rid);
    min_n = std::max<u32>((u32)P.min_cnt, (u32)div_up((u64)P.min_sc, P.hpc ? 255 : (u64)P.k));
    cp.max_dist_x = std::max(P.max_gap, P.bw); cp.max_dist_y = std::max(P.max_gap, P.bw);
    cp.bw = P.bw; cp.max_skip = P.max_skip; cp.max_iter = P.max_iter; cp.min_cnt = P.min_cnt; cp.min_sc = P.min_sc;
    cp.max_drop = P.bw; cp.pen_gap = P.pen_gap; cp.pen_skip = P.pen_skip;
    cp.remove_internal = job.prm.remove_internal ? (job.mode == MODE_INVERSE ? 2 : 1) : 0;
    cp.max_overhang_ratio = job.prm.max_overhang_ratio;
    cp.want_all = (job.n_chains != nullptr || cp.remove_internal) ? 1 : 0;
    cp.q_len = Q->d_len; cp.t_len = T->d_len;
    return LRGE_OK;
}

// The expansion emits the anchors query by query, so only (target, strand, position) need sorting, inside every
// query's segment.  Packed (count-only) runs sort the segments that fit a workgroup's LDS there (k_seg_sort_local,
// capacity classes 2048 / 8192 / 16384 anchors); everything else is cut into RS_TILE tiles for the segmented
// global passes (SegTile, k_prims.h), whose scanned histogram is offset by the items sorted locally (delta).
// kept != null (dead-pair filter): query q's anchors are the first kept[q - q0] of its slot of h_qtot[q] in the expansion's
// output; the sort gathers them from there (src) into the dense layout (start) every later stage works in.
void OverlapRun::plan_anchor_sort(u32 q0, u32 q1, bool packed, const u32 *kept) {
    h_tiles.clear();
    for (auto &v : h_local) v.clear();
    u32 off = 0, src = 0, tb = 0, &n_local = n_local_items;
    n_local = 0;
    const bool local_ok = !ctx->opt("NO_LOCAL_SORT");
    const int local_max = ctx->opt("LOCAL_SORT_MAX") ? atoi(ctx->opt("LOCAL_SORT_MAX")) : 2;   // largest class sorted in LDS
    for (u32 q = q0; q < q1; ++q) {
        const u32 c = kept ? kept[q - q0] : h_qtot[q], slot = h_qtot[q];
        if (packed && c) {
            const int cls = c <= 2048 ? 0 : c <= 8192 ? 1 : c <= 16384 ? 2 : 3;
            if (cls < 3 && cls <= local_max && local_ok && ctx->lsort_ok[cls]) { h_local[cls].push_back(SegDesc{off, c, q - q0, src}); off += c; src += slot; n_local += c; continue; }
        }
        const u32 nt_q = (u32)div_up((u64)c, RS_TILE);
        for (u32 lt = 0; lt < nt_q; ++lt) {
            SegTile t; t.start = off + lt * RS_TILE; t.len = std::min<u32>(RS_TILE, c - lt * RS_TILE);
            t.hbase = 256u * tb + lt; t.hstride = nt_q; t.seg = q - q0; t.delta = n_local; t.src = src + lt * RS_TILE; t.pad = 0;
            h_tiles.push_back(t);
        }
        off += c; src += slot; tb += nt_q;
    }
}

int OverlapRun::dump_sorted_anchors(const u64 *skey, const u64 *sval, u64 A) {
    *job.an = A;
    u64 m = A < job.acap ? A : job.acap;
    std::vector<u64> hk(m), hvv(m);
    if (m) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
        HIPCHK(ctx, hipMemcpy(hk.data(), skey, m * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(hvv.data(), sval, m * 8, hipMemcpyDeviceToHost));
    }
    const u64 rmask = (1ULL << kl.bits_rpos) - 1;
    // back to minimap2's mm128 anchor encoding and array order: the device orders groups
    // (target, strand) so that both strands of a pair are adjacent, minimap2 orders them
    // (strand, target); a stable re-sort by x keeps the order inside every group.
    std::vector<std::pair<u64, u64>> tmp(m);
    for (u64 i = 0; i < m; ++i) {
        u64 k = hk[i];
        u64 rev = (k >> kl.sh_rev()) & 1, rid = (k >> kl.sh_rid()) & ((1ULL << kl.bits_rid) - 1);
        tmp[i] = {rev << 63 | rid << 32 | (k & rmask), hvv[i] & AVAL_LOW_MASK};   // drop the seed rank
    }
    std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<u64, u64> &a, const std::pair<u64, u64> &b) { return a.first < b.first; });
    for (u64 i = 0; i < m; ++i) { job.ax[i] = tmp[i].first; job.ay[i] = tmp[i].second; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RUN_DONE;
}

int OverlapRun::batch(u32 q0, u32 q1, u64 A) {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    int rc = 0;
    const u64 mb = h_mzoff[q0], me = h_mzoff[q1];
    if (A == 0 || me == mb) return LRGE_OK;
    ctx->counters[LRGE_C_ANCHORS] += A;
    Scratch bsc(ctx);
    bsc.max_bytes = (size_t)ctx->opt_u64("DEBUG_BATCH_ALLOC_MAX_BYTES", 0);
    u64 *akey, *aval, *akey2, *aval2, *skey, *sval;
    // count-only runs carry one packed u64 per anchor through the expansion and the sort (k_prims.h UnpackParams);
    // chain records (PAF) need the seed rank as well and keep the (key, value) pairs
    const u32 bits_qy = std::max<u32>(1, ceil_log2_u64((u64)Q->max_len + 1));
    const bool packed = !d_chains && !job.dump_anchors && kl.sh_q() + bits_qy + 9 <= 64 && !ctx->opt_u64("NO_PACKED", 0);
    // Dead-pair filter (count-only runs, k_seed.h: k_expand_q): anchors of (target, strand) pairs that cannot reach the min_n anchors
    // the group stage asks for are dropped where they are made -- A shrinks to what the sort, the group stage and the chain
    // kernels see (a third of it at H. sapiens scale).  option NO_GROUP_FILTER: the plain expansion (tests compare the two).
    const bool filt = packed && !ctx->opt("NO_GROUP_FILTER") && min_n >= 2;
    const u64 A_all = A;
    {
        StageTimer t(ctx, LRGE_T_EXPAND);
        const u32 *aoff = aoff_all;
        if (!aoff) return LRGE_ERR_DEVICE;
        if (filt) {
            akey = bsc.get<u64>(A_all + 8);
            const u32 nqb = q1 - q0;
            u32 *d_kept = bsc.get<u32>((size_t)nqb + 1), *d_qlist = bsc.get<u32>((size_t)nqb + 1);
            if (!akey || !d_kept || !d_qlist) return LRGE_ERR_DEVICE;
            // the queries by the size of their slot: one launch per class (k_seed.h), largest first so that the long workgroups start early
            h_qlist.resize(nqb);
            u32 n_cls[3] = {0, 0, 0};
            const u32 smax = (u32)ctx->opt_u64("DEBUG_EXPQ_SMALL_MAX", EXPQ_SMALL_MAX), mmax = (u32)ctx->opt_u64("DEBUG_EXPQ_MID_MAX", EXPQ_MID_MAX);   // (tests: every class on small sets)
            auto cls_of = [&](u32 c) { return c <= smax ? 0 : c <= mmax ? 1 : 2; };
            for (u32 q = q0; q < q1; ++q) ++n_cls[cls_of(h_qtot[q])];
            u32 at[3] = {n_cls[2] + n_cls[1], n_cls[2], 0};
            for (u32 q = q0; q < q1; ++q) h_qlist[at[cls_of(h_qtot[q])]++] = q - q0;
            HIPCHK(ctx, hipMemcpyAsync(d_qlist, h_qlist.data(), (size_t)nqb * 4, hipMemcpyHostToDevice, ctx->stream));
            const u32 npl = std::min<u32>(min_n, EXPQ_PLANES);
            if (n_cls[2]) hipLaunchKernelGGL((k_expand_q<1024, 17>), dim3(n_cls[2]), dim3(1024), 0, ctx->stream, so.x, so.y, mb, sp, hs, hn, aoff, so.mz_off, q0, d_qlist, kl, akey, bits_qy, npl, d_kept);
            if (n_cls[1]) hipLaunchKernelGGL((k_expand_q<512, 16>), dim3(n_cls[1]), dim3(512), 0, ctx->stream, so.x, so.y, mb, sp, hs, hn, aoff, so.mz_off, q0, d_qlist + n_cls[2], kl, akey, bits_qy, npl, d_kept);
            if (n_cls[0]) hipLaunchKernelGGL((k_expand_q<256, 14>), dim3(n_cls[0]), dim3(256), 0, ctx->stream, so.x, so.y, mb, sp, hs, hn, aoff, so.mz_off, q0, d_qlist + n_cls[2] + n_cls[1], kl, akey, bits_qy, npl, d_kept);
            KCHK(ctx);
            t.stop();
            h_qkept.resize((size_t)(q1 - q0));
            HIPCHK(ctx, ctx->d2h(h_qkept.data(), d_kept, (size_t)(q1 - q0) * 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            bsc.drop(d_kept); bsc.drop(d_qlist);
            A = 0;
            for (u32 c : h_qkept) A += c;
            ctx->counters[LRGE_C_ANCHORS_KEPT] += A;
            if (A == 0) return LRGE_OK;                       // nothing can chain: every count of the batch stays 0
            // (+8: k_chain_lpg streams anchors in 16-byte pairs and may read one element past the last group)
            aval = bsc.get<u64>(A + 8); akey2 = bsc.get<u64>(A + 8); aval2 = bsc.get<u64>(A + 8);
            if (!aval || !akey2 || !aval2) return LRGE_ERR_DEVICE;
        } else {
            // (+8: k_chain_lpg streams anchors in 16-byte pairs and may read one element past the last group)
            akey = bsc.get<u64>(A + 8); aval = bsc.get<u64>(A + 8); akey2 = bsc.get<u64>(A + 8); aval2 = bsc.get<u64>(A + 8);
            if (!akey || !aval || !akey2 || !aval2) return LRGE_ERR_DEVICE;
            hipLaunchKernelGGL(k_expand, dim3((u32)div_up(me - mb, 256)), dim3(256), 0, ctx->stream, so.x, so.y, mb, me, sp, hs, hn, aoff,
                               need_rank ? krank : (const u32 *)nullptr, so.mz_off, q0, kl, akey, aval, packed ? bits_qy : 0u);
            KCHK(ctx);
            ctx->counters[LRGE_C_ANCHORS_KEPT] += A;
            // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
            t.stop();
        }
    }
    {
        StageTimer t(ctx, LRGE_T_ANCHOR_SORT);
        plan_anchor_sort(q0, q1, packed, filt ? h_qkept.data() : nullptr);
        SegTile *d_tiles = (SegTile *)bsc.get<u32>(h_tiles.size() * (sizeof(SegTile) / 4) + 4);
        if (!d_tiles) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(d_tiles, h_tiles.data(), h_tiles.size() * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
        if (packed) {
            UnpackParams up; up.sb = kl.sh_q(); up.bits_qy = bits_qy; up.sh_q = kl.sh_q(); up.dmask = 255;
            // segments that fit a workgroup's LDS are sorted there in one kernel (k_seg_sort_local: 8 B in, 16 B out
            // per anchor); only the larger ones take the tiled global passes
            // the classes touch disjoint segments: the largest class runs on the side stream beside the others and the
            // tiled passes (fork / join with events), so that its one-block-per-CU tail does not stand alone
            const bool side = !h_local[2].empty() && (!h_local[1].empty() || !h_tiles.empty()) && !ctx->opt("LSORT_SERIAL");
            SegDesc *d_seg[3] = {nullptr, nullptr, nullptr};
            for (int cls = 0; cls < 3; ++cls) {
                if (h_local[cls].empty()) continue;
                d_seg[cls] = (SegDesc *)bsc.get<u32>(h_local[cls].size() * 4);
                if (!d_seg[cls]) return LRGE_ERR_DEVICE;
                HIPCHK(ctx, hipMemcpyAsync(d_seg[cls], h_local[cls].data(), h_local[cls].size() * sizeof(SegDesc), hipMemcpyHostToDevice, ctx->stream));
            }
            if (side) {
                HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            }
            const int nbits = (int)kl.sh_q();
            if (d_seg[2]) {
                hipLaunchKernelGGL((k_seg_sort_local<1024, 16, LSORT_DB>), dim3((u32)h_local[2].size()), dim3(1024), LSORT_BYTES(1024, 16, LSORT_DB), side ? ctx->stream2 : ctx->stream,
                                   akey, aval, aval2, d_seg[2], up, nbits);
                KCHK(ctx);
                if (side) HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            }
            if (d_seg[1]) {
                hipLaunchKernelGGL((k_seg_sort_local<512, 16, LSORT_DB>), dim3((u32)h_local[1].size()), dim3(512), LSORT_BYTES(512, 16, LSORT_DB), ctx->stream, akey, aval, aval2, d_seg[1], up, nbits);
                KCHK(ctx);
            }
            if (d_seg[0]) {
                hipLaunchKernelGGL((k_seg_sort_local<256, 8, 8>), dim3((u32)h_local[0].size()), dim3(256), LSORT_BYTES(256, 8, 8), ctx->stream, akey, aval, aval2, d_seg[0], up, nbits);
                KCHK(ctx);
            }
            // With the dead-pair filter the segments are READ in the expansion's sparse layout and WRITTEN in the dense one, and the
            // tiled sort's second pass writes into akey -- dense positions that are other queries' unread sparse slots.  The local
            // sorts on this stream are over by then (stream order); the largest class on the side stream is not: the tiled passes
            // wait for it.  (Found at C5/2: counts off on the ~9 000 queries whose slots a tiled segment's output overwrote.)
            bool joined = false;
            if (side && filt && !h_tiles.empty()) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); joined = true; }
            rc = radix_sort_packed_seg(ctx, bsc, akey, akey2, aval, aval2, A, (int)kl.sh_q(), d_tiles, (u32)h_tiles.size(), up, A - n_local_items, /*src_first=*/filt);
            if (rc) return rc;
            if (side && !joined) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            skey = aval; sval = aval2;
            bsc.drop((u32 *)d_tiles);
            bsc.drop(akey); bsc.drop(akey2);
        } else {
            rc = radix_sort_pairs(ctx, bsc, akey, aval, akey2, aval2, A, 0, (int)(kl.bits_rpos + 1 + kl.bits_rid), &skey, &sval, false,
                                  d_tiles, (u32)h_tiles.size());
            if (rc) return rc;
            bsc.drop((u32 *)d_tiles);
            // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
            bsc.drop(skey == akey ? akey2 : akey);
            bsc.drop(sval == aval ? aval2 : aval);
        }
        t.stop();
    }
    if (job.dump_anchors) return dump_sorted_anchors(skey, sval, A);
    // groups.  The size-sorted list of the groups worth chaining is split: groups above lpg_max anchors go to k_chain_hw
    // (short latency per anchor), the rest to k_chain_lpg (64 groups per wavefront).  The split is chosen per batch from
    // the size census of the groups (see below); option LPG_MAX pins it, CHAIN=hw|lpg forces one kernel.
    const char *cm = ctx->opt("CHAIN");
    u32 lpg_max = LPG_MAX_AUTO;
    if (const char *e = ctx->opt("LPG_MAX")) lpg_max = (u32)strtoul(e, nullptr, 10);
    if (cm && !strcmp(cm, "hw")) lpg_max = 0;
    if (cm && !strcmp(cm, "lpg")) lpg_max = 0xFFFFFFFFu;
    if (lpg_max && (cp.want_all || d_chains) && !(cm && !strcmp(cm, "lpg"))) lpg_max = 0;   // records: wave-wide backtrack anyway
    if (cp.max_iter < LPG_W) lpg_max = 0;    // (debug knob only) k_chain_lpg assumes every window slot is a candidate
    u32 n_big = 0, lpg_split = 0;
    u32 G = 0; u32 *gstart, *gflags, *hw_list = nullptr;
    u32 n_chained = 0; unsigned long long a_chained = 0, a_big = 0;
    {
        StageTimer t(ctx, LRGE_T_GROUP);
        u32 *d_G = bsc.get<u32>(1);
        {
            // group starts into an upper-bound block (one entry per anchor): the group count stays on the device
            // until it travels to the host together with the size census -- one round trip instead of two
            gstart = bsc.get<u32>((size_t)A + 1);
            if (!gstart || !d_G) return LRGE_ERR_DEVICE;
            rc = compact_heads_async(ctx, bsc, skey, A, kl.bits_rpos, gstart, d_G);   // runs of equal (query, target, strand)
            if (rc) return rc;
        }
        {
            // groups worth chaining, sorted by size (largest first) so that k_chain_hw pairs equals
            u32 *d_cnt = bsc.get<u32>(4 + GSZ_BINS);
            unsigned long long *d_anch = (unsigned long long *)bsc.get<u64>(2 + GSZ_BINS);
            if (!d_cnt || !d_anch) return LRGE_ERR_DEVICE;
            HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, (4 + GSZ_BINS) * 4, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(d_anch, 0, (2 + GSZ_BINS) * 8, ctx->stream));
            hipLaunchKernelGGL(k_group_count, dim3((u32)std::min<u64>(div_up(A, 4096), (u64)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, gstart, d_G, A, min_n,
                               d_cnt, d_anch, d_cnt + 4, d_anch + 2);
            KCHK(ctx);
            u32 h_cnt[4 + GSZ_BINS]; unsigned long long h_anch[2 + GSZ_BINS];
            HIPCHK(ctx, ctx->d2h(&G, d_G, 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h(h_cnt, d_cnt, sizeof(h_cnt), ctx->stream));
            HIPCHK(ctx, ctx->d2h(h_anch, d_anch, sizeof(h_anch), ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            gflags = bsc.get<u32>((size_t)G + 1);
            if (!gflags) return LRGE_ERR_DEVICE;
            HIPCHK(ctx, hipMemsetAsync(gflags, 0, ((size_t)G + 1) * 4, ctx->stream));
            n_chained = h_cnt[0]; a_chained = h_anch[0];
            {
                const ChainSplit sp_ = choose_chain_split(h_cnt + 4, h_anch + 2, a_chained, lpg_max, ctx->n_cu);
                n_big = sp_.n_big; a_big = sp_.a_big; lpg_split = sp_.T;
                ctx->counters[LRGE_C_LPG_SPLIT] = lpg_split;
                if (ctx->opt("VERBOSE"))
                    fprintf(stderr, "[lrge_hip] batch: %u groups chained, %llu anchors, largest class %d (<= %d anchors), split T=%u -> hw %u groups / %llu anchors\n",
                            n_chained, a_chained, sp_.top, (sp_.top + 1) * GSZ_W, sp_.T, n_big, a_big);
            }
            if (n_chained) {
                u64 *k0 = bsc.get<u64>(n_chained), *v0 = bsc.get<u64>(n_chained), *k1 = bsc.get<u64>(n_chained), *v1 = bsc.get<u64>(n_chained);
                hw_list = bsc.get<u32>(n_chained);
                if (!k0 || !v0 || !k1 || !v1 || !hw_list) return LRGE_ERR_DEVICE;
                hipLaunchKernelGGL(k_group_fill, dim3((u32)div_up(G, GB_CHUNK)), dim3(256), 0, ctx->stream, gstart, G, A, min_n, d_cnt + 1, k0, v0);
                KCHK(ctx);
                u64 *rk, *rv;
                rc = radix_sort_pairs(ctx, bsc, k0, v0, k1, v1, n_chained, 0, 16, &rk, &rv);   // keys: 65535 - min(n, 65535)
                if (rc) return rc;
                hipLaunchKernelGGL(k_vals_to_u32, dim3((u32)div_up(n_chained, 256)), dim3(256), 0, ctx->stream, rv, n_chained, hw_list);
                KCHK(ctx);
                bsc.drop(k0); bsc.drop(v0); bsc.drop(k1); bsc.drop(v1);
            }
            bsc.drop(d_cnt); bsc.drop(d_anch);
        }
        t.stop();
    }
    ctx->counters[LRGE_C_GROUPS] += G;
    {
        GroupOut go; go.flags = gflags; go.chains = d_chains; go.n_chains = d_nchains; go.chain_cap = job.chain_cap; go.rid_base = job.rid_base;
        {
            if (n_chained) {
                StageTimer t(ctx, LRGE_T_CHAIN);
                HwChainArgs ha;
                ha.akey = skey; ha.aval = sval; ha.gstart = gstart; ha.n_groups = G; ha.n_anchors = A; ha.list = hw_list; ha.n_list = n_big;
                ha.grec = bsc.get<u64>(A); ha.tmark = bsc.get<u32>(A);
                ha.prio = (u32)ctx->opt_u64("HW_PRIO", 0);
                if (!ha.grec || !ha.tmark) return LRGE_ERR_DEVICE;
                HIPCHK(ctx, hipMemsetAsync(ha.tmark, 0, A * 4, ctx->stream));
                // the list is sorted by min(n, 65535) descending, so [0, n_big) are exactly the groups above lpg_max
                // the two kernels touch disjoint groups; k_chain_lpg goes to the side stream so that its long
                // wavefronts run beside k_chain_hw's (fork / join with events, no host sync)
                const bool both = n_big && n_chained > n_big;
                if (both) {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                }
                if (n_chained > n_big) {
                    LpgChainArgs la;
                    la.akey = skey; la.aval = sval; la.gstart = gstart; la.n_groups = G; la.n_anchors = A;
                    la.list = hw_list + n_big; la.n_list = n_chained - n_big; la.grec = ha.grec; la.tmark = ha.tmark;
                    la.prio = (u32)ctx->opt_u64("LPG_PRIO", 3);
                    // 1024: on clean input (C2) no group is given up -- redoing even one 500-anchor group costs 0.3 ms of
                    // critical path; on a repeat-rich genome (synth c2_repeats) 64 would be ~1.7x faster still
                    la.slow_budget = (u32)ctx->opt_u64("LPG_SLOW_BUDGET", 1024);
                    la.slow_entries = (u32)ctx->opt_u64("LPG_SLOW_ENTRIES", 4);
                    la.no_prune = ctx->opt("LPG_NO_PRUNE") ? 1u : 0u;
                    la.redo_list = bsc.get<u32>((size_t)la.n_list + 1); la.redo_count = bsc.get<u32>(1);
                    if (!la.redo_list || !la.redo_count) return LRGE_ERR_DEVICE;
                    HIPCHK(ctx, hipMemsetAsync(la.redo_count, 0, 4, both ? ctx->stream2 : ctx->stream));
                    StageTimer tl(ctx, LRGE_T_CHAIN_LPG, both ? ctx->stream2 : ctx->stream);
                    const bool pentab = cp.pen_skip == 0.0f && cp.bw >= 0 && cp.bw + 2 <= 8192 && !ctx->opt("LPG_NOTAB");
                    const bool fastreach = cp.max_iter >= 64 && !ctx->opt("LPG_EXACT_REACH");
                    const dim3 lgrid((la.n_list + 64 * LPG_WAVES - 1) / (64 * LPG_WAVES)), lblock(64 * LPG_WAVES);
                    const size_t lds_ring = (size_t)LPG_WAVES * LPG_RING_BYTES;
                    const size_t lds_tab = (((size_t)cp.bw + 2) * 4 + 15) / 16 * 16 + lds_ring;
                    hipStream_t lst = both ? ctx->stream2 : ctx->stream;
                    if (pentab && fastreach) hipLaunchKernelGGL((k_chain_lpg<true, true>), lgrid, lblock, lds_tab, lst, la, cp, go);
                    else if (pentab) hipLaunchKernelGGL((k_chain_lpg<true, false>), lgrid, lblock, lds_tab, lst, la, cp, go);
                    else if (fastreach) hipLaunchKernelGGL((k_chain_lpg<false, true>), lgrid, lblock, lds_ring, lst, la, cp, go);
                    else hipLaunchKernelGGL((k_chain_lpg<false, false>), lgrid, lblock, lds_ring, lst, la, cp, go);
                    KCHK(ctx);
                    tl.stop();
                    ctx->counters[LRGE_C_CHAIN_LAUNCHES] += 1;
                    ctx->counters[LRGE_C_LPG_LAUNCHES] += 1;
                    ctx->counters[LRGE_C_LPG_ANCHORS] += a_chained - a_big;
                    {   // the groups k_chain_lpg gave up (slow-path budget), on the same stream right behind it -- beside
                        // k_chain_hw's tail.  Usually none: then this is an empty launch.  Their number only exists on the
                        // device: as many wavefronts as the chip holds stride the list.
                        HwChainArgs hr = ha;
                        hr.list = la.redo_list; hr.n_list = 0; hr.prio = 0;
                        const u32 redo_grid = (u32)std::min<u64>(((u64)la.n_list + 1) / 2, (u64)ctx->n_cu * 32);
                        hipLaunchKernelGGL(k_chain_hw_redo, dim3(std::max<u32>(redo_grid, 1)), dim3(64), 0, both ? ctx->stream2 : ctx->stream, hr, cp, go, la.redo_count);
                        KCHK(ctx);
                        if (ctx->opt("VERBOSE")) {
                            u32 nr = 0;
                            HIPCHK(ctx, hipMemcpyAsync(&nr, la.redo_count, 4, hipMemcpyDeviceToHost, both ? ctx->stream2 : ctx->stream));
                            HIPCHK(ctx, hipStreamSynchronize(both ? ctx->stream2 : ctx->stream));
                            fprintf(stderr, "[lrge_hip] k_chain_lpg handed %u of %u groups to k_chain_hw_redo\n", nr, la.n_list);
                        }
                    }
                }
                if (n_big) {
                    hipLaunchKernelGGL(k_chain_hw, dim3((n_big + 1) / 2), dim3(64), 0, ctx->stream, ha, cp, go);
                    KCHK(ctx);
                    ctx->counters[LRGE_C_CHAIN_LAUNCHES] += 1;
                }
                if (both) {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                }
                t.stop();
                ctx->counters[LRGE_C_CHAIN_ANCHORS] += a_chained;
                ctx->counters[LRGE_C_GROUPS_CHAINED] += n_chained;
            }
        }
    }
    {
        StageTimer t(ctx, LRGE_T_COUNT);
        CountParams cnp; cnp.kl = kl; cnp.q0 = q0; cnp.mode = job.mode;
        cnp.q_rank = Q->has_rank ? Q->d_rank : nullptr; cnp.t_rank = T->has_rank ? T->d_rank : nullptr;
        cnp.t_dup = T->dup_rank ? 1 : 0;
        cnp.q_map = d_qmap; cnp.rid_base = job.rid_base;
        if (n_chained) {
            hipLaunchKernelGGL(k_count, dim3((u32)div_up(n_chained, 256)), dim3(256), 0, ctx->stream, skey, gstart, gflags, hw_list, n_chained, cnp, d_counts, d_hasmap);
            KCHK(ctx);
        }
        // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
        t.stop();
    }
    return LRGE_OK;
}

int OverlapRun::finish() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    if (job.counts) HIPCHK(ctx, ctx->d2h(job.counts, d_counts, (size_t)n_out * 4, ctx->stream));
    if (job.has_map) HIPCHK(ctx, ctx->d2h(job.has_map, d_hasmap, (size_t)nq * 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    if (job.n_chains) {
        unsigned long long nchn = 0;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
        HIPCHK(ctx, hipMemcpy(&nchn, d_nchains, 8, hipMemcpyDeviceToHost));
        *job.n_chains = nchn;
        u64 m = nchn < job.chain_cap ? nchn : job.chain_cap;
        if (m && job.chains) HIPCHK(ctx, hipMemcpy(job.chains, d_chains, m * sizeof(lrge_hip_chain), hipMemcpyDeviceToHost));
    }
    if (job.an && job.dump_anchors) *job.an = 0;
    return LRGE_OK;
}

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
            ctx->counters[LRGE_C_BATCHES] -= 1; ctx->counters[LRGE_C_ANCHORS] -= A;
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
    float ms[LRGE_T_N]; u64 cn[LRGE_C_N];
    StageAcc() { memset(ms, 0, sizeof ms); memset(cn, 0, sizeof cn); }
    void add(const lrge_hip_ctx *ctx) {
        for (int i = 0; i < LRGE_T_N; ++i) ms[i] += ctx->ms[i];
        for (int i = 0; i < LRGE_C_N; ++i) cn[i] = i == LRGE_C_LPG_SPLIT ? ctx->counters[i] : cn[i] + ctx->counters[i];
    }
    void store(lrge_hip_ctx *ctx) const { memcpy(ctx->ms, ms, sizeof ms); memcpy(ctx->counters, cn, sizeof cn); }
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
    // parts and it has a mapping if it has one in any part; every part sees the same queries and the global mid_occ
    const u32 nq = queries->n;
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
