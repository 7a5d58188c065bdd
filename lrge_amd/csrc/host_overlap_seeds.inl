// host_overlap_seeds.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): the overlap core, first half: OverlapJob / OverlapRun, the chain-split chooser, and the stages in front of the batches -- prepare (outputs, shard map), seeds (K1 sketch, K3 lookup, K4a query-occurrence filter, hit counts), plan (batch size, key layout, chaining parameters).
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
    kl.bits_rpos = std::max<u32>(1, ceil_log2_u64((u64)T->max_len + 1));
    kl.bits_rid = std::max<u32>(1, ceil_log2_u64((u64)nt));
    max_bits_q = 63 - (kl.bits_rpos + 1 + kl.bits_rid);
    min_n = std::max<u32>((u32)P.min_cnt, (u32)div_up((u64)P.min_sc, P.hpc ? 255 : (u64)P.k));
    // With the dead-pair filter (count-only runs: batch(), k_expand_q) a batch is counted in seed HITS, of which only the survivors --
    // a quarter at H. sapiens scale, under half on noisy ONT reads -- go through the sort and the chain kernels: the 32-bit positions
    // that bound a batch are then the hits' slots (offsets relative to the batch: anything below 2^32), and a hit costs its 8-byte slot
    // plus 40 B per SURVIVOR.  The survivors' share is what this context's recent batches showed (ctx->kept_ratio, + 15 % + 0.02; half
    // of the hits before the first batch has been seen): 21-24 B per hit at H. sapiens scale.  Full-size C5 (round 5): one batch per
    // index part under ava-pb instead of two (3.6 G hits, 1.2 G kept), two instead of three under ava-ont.  A batch whose survivors do
    // not fit after all is retried in halves (run_overlap).
    const u32 bits_qy_ = std::max<u32>(1, ceil_log2_u64((u64)Q->max_len + 1));
    const bool filt_expected = !d_chains && !job.dump_anchors && kl.sh_q() + bits_qy_ + 9 <= 64 && !ctx->opt_u64("NO_PACKED", 0) &&
                               !ctx->opt("NO_GROUP_FILTER") && min_n >= 2 && !ctx->opt("BATCH_HITS_2G");
    const u64 per_item = filt_expected ? 8 + (u64)std::ceil(40.0 * std::min(1.0, ctx->kept_ratio * 1.15 + 0.02)) : 48;
    batch_cap = filt_expected ? (1ULL << 32) - (1ULL << 26) : 1ULL << 31;
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
            const u64 by_mem = ((u64)mfree + ctx->pool.idle()) / 5 * 4 / per_item;     // the pool's idle blocks are reusable too (not the ones in use: a resident index)
            if (by_mem < batch_cap) batch_cap = by_mem;
        }
        if (batch_cap < (1ULL << 20)) batch_cap = 1ULL << 20;
    }
    batch_cap = ctx->opt_u64("BATCH_ANCHORS", batch_cap);
    cp.max_dist_x = std::max(P.max_gap, P.bw); cp.max_dist_y = std::max(P.max_gap, P.bw);
    cp.bw = P.bw; cp.max_skip = P.max_skip; cp.max_iter = P.max_iter; cp.min_cnt = P.min_cnt; cp.min_sc = P.min_sc;
    cp.max_drop = P.bw; cp.pen_gap = P.pen_gap; cp.pen_skip = P.pen_skip;
    cp.remove_internal = job.prm.remove_internal ? (job.mode == MODE_INVERSE ? 2 : 1) : 0;
    cp.max_overhang_ratio = job.prm.max_overhang_ratio;
    cp.want_all = (job.n_chains != nullptr || cp.remove_internal) ? 1 : 0;
    cp.q_len = Q->d_len; cp.t_len = T->d_len;
    return LRGE_OK;
}

