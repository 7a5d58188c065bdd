// host_tshard.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): lrge_hip_index_build_tsharded, the
// collective index build of the forward strategy with the TARGETS sharded over the ranks (k_tshard.h says why and what travels).
// ------------------------------------------------------------------------------------------
// A collective call: the sequence of collectives is fixed, a rank that fails joins the next one in its own shape with the status
// word set (CollectiveGuard, host_index_collective.inl).
//   C1 all-gather  u64[W + 1]   pairs this rank sends to every owner, status
//   A1 agreement                (send / receive buffers taken)
//   C2 all-to-all  u64          the pairs (hash << 24 | local count)
//   C4 all-reduce  u64[head+3]  distinct keys, minimizers, head bins of the occurrence histogram, status   -> mid_occ
//   C5 all-gather  u64[2]       too-frequent keys this rank owns, status
//   A2 agreement                (room for everybody's list)
//   C6 all-gather  u64[max]     the lists
//   A3 agreement                (the lists arrived and the tables are marked on every rank: nobody leaves with LRGE_OK alone)
struct TsTable { u64 *ht; u64 cap, slots; };

static int ts_global_stats(lrge_hip_ctx *ctx, lrge_hip_index *ix, lrge_hip_comm *c, CollectiveGuard &cg) {
    const int W = c->world, me = c->rank;
    hipStream_t st = ctx->stream;
    const Preset &P = ix->P;
    Scratch sc(ctx);
    std::vector<TsTable> tabs;
    if (ix->parts.empty()) tabs.push_back(TsTable{ix->d_ht, ix->ht_cap, ix->ht_slots});
    else for (lrge_hip_index *p : ix->parts) tabs.push_back(TsTable{p->d_ht, p->ht_cap, p->ht_slots});
    const u32 fix = ix->parts.empty() ? ix->ht_fix : ix->parts[0]->ht_fix;
    u64 local_mz = ix->parts.empty() ? ix->n_mz : 0;
    if (!ix->parts.empty()) for (lrge_hip_index *p : ix->parts) local_mz += p->n_mz;
    (void)local_mz;
    // option VERBOSE: time this rank spent in each phase, the waits for the other ranks (local transport) taken out
    double t_mark = DevPool::now_ms(), w_mark = c->wait_ms;
    auto mark = [&](const char *what) {
        if (!ctx->opt("VERBOSE")) return;
        (void)hipStreamSynchronize(st);
        const double now = DevPool::now_ms();
        fprintf(stderr, "[lrge_hip] rank %d target-sharded build: %-28s %7.3f ms (+ %.3f ms waiting)\n", me, what, (now - t_mark) - (c->wait_ms - w_mark), c->wait_ms - w_mark);
        t_mark = now; w_mark = c->wait_ms;
    };
    // ---- C1: how many (key, count) pairs go to every owner ----
    std::vector<u64> mine((size_t)W + 1, 0), matrix(((size_t)W + 1) * (size_t)W, 0);
    unsigned long long *d_tot = nullptr;
    auto local1 = [&]() -> int {
        if (shard_fail_at(ctx, 11)) return LRGE_ERR_DEVICE;
        d_tot = (unsigned long long *)sc.get<u64>(2 * TS_MAX_WORLD);
        if (!d_tot) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemsetAsync(d_tot, 0, 2 * TS_MAX_WORLD * 8, st));
        for (const TsTable &t : tabs)
            if (t.slots) { hipLaunchKernelGGL(k_ts_count, dim3((u32)std::min<u64>(div_up(t.slots, TS_THREADS), (u64)ctx->n_cu * 16)), dim3(TS_THREADS), 0, st, t.ht, t.slots, (u32)W, d_tot); KCHK(ctx); }
        std::vector<unsigned long long> h(TS_MAX_WORLD);
        HIPCHK(ctx, hipMemcpyAsync(h.data(), d_tot, TS_MAX_WORLD * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int o = 0; o < W; ++o) mine[(size_t)o] = h[(size_t)o];
        return LRGE_OK;
    };
    int rc = local1();
    mark("pair counts");
    const bool failed1 = rc != LRGE_OK;
    mine[(size_t)W] = failed1 ? 1 : 0;
    cg.disarm();
    rc = comm_allgather_host(c, mine.data(), mine.size() * 8, matrix.data(), st); if (rc) return rc;
    const size_t row = (size_t)W + 1;
    for (int r = 0; r < W; ++r) if (matrix[(size_t)r * row + W]) { if (!failed1) LRGE_SET_ERR(ctx, "target-sharded index build: rank %d failed", r); return LRGE_ERR_DEVICE; }
    std::vector<u64> s_off((size_t)W + 1, 0), r_off((size_t)W + 1, 0);
    for (int d = 0; d < W; ++d) { s_off[(size_t)d + 1] = s_off[(size_t)d] + mine[(size_t)d]; r_off[(size_t)d + 1] = r_off[(size_t)d] + matrix[(size_t)d * row + (size_t)me]; }
    const u64 n_s = s_off[(size_t)W], n_r = r_off[(size_t)W];
    // (hashes_sent / _recv slots: pairs of 8 bytes; the minimizers a sharded presketch of this step's streamed set moved stay in the entries slots)
    const bool qs = ctx->qshard_fresh; ctx->qshard_fresh = false;
    const u64 ss[8] = {0, 0, qs ? ctx->shard_stats[2] : 0, qs ? ctx->shard_stats[3] : 0, n_s - mine[(size_t)me], n_r - mine[(size_t)me], (u64)(qs ? (ctx->shard_stats[6] & 0xff) : 8) | (u64)8 << 8, 0};
    memcpy(ctx->shard_stats, ss, sizeof ss);
    // ---- A1, C2, C3: the pairs travel ----
    u64 *sh = nullptr, *rh = nullptr;
    cg.expect(CollectiveGuard::AGREE);
    auto local2 = [&]() -> int {
        if (shard_fail_at(ctx, 12)) return LRGE_ERR_DEVICE;
        if (n_r >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "target-sharded index build: this rank owns %llu (key, count) pairs (limit 2^32)", (unsigned long long)n_r); return LRGE_ERR_TOO_MANY; }
        sh = sc.get<u64>(n_s + 1); rh = sc.get<u64>(n_r + 1);
        if (!sh || !rh) return LRGE_ERR_DEVICE;
        std::vector<unsigned long long> cur(TS_MAX_WORLD, 0);
        for (int o = 0; o < W; ++o) cur[(size_t)o] = s_off[(size_t)o];
        HIPCHK(ctx, hipMemcpyAsync(d_tot + TS_MAX_WORLD, cur.data(), TS_MAX_WORLD * 8, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));          // (`cur` is a local)
        for (const TsTable &t : tabs)
            if (t.slots) { hipLaunchKernelGGL(k_ts_emit, dim3((u32)div_up(t.slots, TS_THREADS * TS_ITEMS)), dim3(TS_THREADS), 0, st, t.ht, t.slots, (u32)W, d_tot + TS_MAX_WORLD, sh); KCHK(ctx); }
        return LRGE_OK;
    };
    rc = local2();
    mark("pair emit");
    cg.disarm();
    rc = comm_agree(c, rc, st); if (rc) return rc;
    rc = comm_alltoallv(c, sh, s_off.data(), rh, r_off.data(), 8, st); if (rc) return rc;
    mark("all-to-alls");
    // ---- C4: the owner adds the counts up; the statistics of the one index ----
    const u32 max_bin = (u32)P.max_mid_occ + 1, head = std::min<u32>(4096, max_bin + 1);
    cg.expect(CollectiveGuard::ALLREDUCE_U64, stats_vec_words(P), stats_vec_words(P) - 1);
    u64 *rk = nullptr; u32 *starts = nullptr, *d_nr = nullptr, *gcnt = nullptr, *d_hist = nullptr;
    unsigned long long *d_mz = nullptr;
    std::vector<u64> hv((size_t)head + 3, 0);
    {
        if (shard_fail_at(ctx, 13)) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipStreamSynchronize(st));        // (the offset vectors above are locals)
        sc.drop(sh);
        ALLOC_OR_FAIL(k1, sc, u64, n_r + 1);
        rc = radix_sort_keys(ctx, sc, rh, k1, n_r, TS_CNT_BITS, 2 * P.k, &rk, /*reverse_digits=*/false); if (rc) return rc;
        sc.drop(rk == rh ? k1 : rh);                  // (the other half of the ping-pong: recycled in stream order)
        u64 *rv = rk;
        starts = sc.get<u32>(n_r + 2); d_nr = sc.get<u32>(1); gcnt = sc.get<u32>(n_r + 1); d_hist = sc.get<u32>((size_t)max_bin + 2);
        d_mz = (unsigned long long *)sc.get<u64>(1);
        if (!starts || !d_nr || !gcnt || !d_hist || !d_mz) return LRGE_ERR_DEVICE;
        rc = compact_heads_async(ctx, sc, rk, n_r, TS_CNT_BITS, starts, d_nr); if (rc) return rc;
        HIPCHK(ctx, hipMemsetAsync(d_hist, 0, ((size_t)max_bin + 2) * 4, st));
        HIPCHK(ctx, hipMemsetAsync(d_mz, 0, 8, st));
        if (n_r) {
            hipLaunchKernelGGL(k_ts_reduce, dim3((u32)std::min<u64>(div_up(n_r, 256), (u64)ctx->n_cu * 8)), dim3(256), 0, st, rv, starts, d_nr, n_r, gcnt, d_hist, max_bin, d_mz);
            KCHK(ctx);
        } else HIPCHK(ctx, hipMemsetAsync(d_nr, 0, 4, st));
        u32 h_nr = 0; unsigned long long h_mz = 0;
        std::vector<u32> hb(head);
        HIPCHK(ctx, hipMemcpyAsync(&h_nr, d_nr, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(&h_mz, d_mz, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(hb.data(), d_hist, (size_t)head * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        hv[0] = h_nr; hv[1] = h_mz;
        for (u32 b = 0; b < head; ++b) hv[2 + (size_t)b] = hb[b];
    }
    mark("owner: sort + reduce");
    cg.disarm();
    rc = comm_allreduce_sum_host(c, hv.data(), hv.size(), 8, st); if (rc) return rc;
    mark("statistics all-reduce");
    if (hv[(size_t)head + 2]) { LRGE_SET_ERR(ctx, "target-sharded index build: %llu other rank(s) failed", (unsigned long long)hv[(size_t)head + 2]); return LRGE_ERR_DEVICE; }
    const u64 g_distinct = hv[0], g_mz = hv[1];
    int thres = INT32_MAX;
    if (g_distinct) {      // mm_idx_cal_max_occ + the clamps of mm_mapopt_update, over the distinct keys of ALL targets (index_build_one's arithmetic)
        const u64 kth = (u64)((1. - (double)P.mid_occ_frac) * (double)g_distinct);
        u64 cum = 0; u32 v = max_bin; bool found = false;
        for (u32 b = 0; b < head; ++b) { cum += hv[2 + (size_t)b]; if (cum > kth) { v = b; found = true; break; } }
        if (!found && head < max_bin + 1) {      // beyond the head bins: the whole histogram travels (every rank takes this branch or none)
            cg.expect(CollectiveGuard::AGREE);
            u64 *d_full = sc.get<u64>((size_t)max_bin + 1);
            int arc = d_full ? LRGE_OK : LRGE_ERR_DEVICE;
            if (d_full) { hipLaunchKernelGGL(k_u32_to_u64, dim3((u32)div_up((u64)max_bin + 1, 256)), dim3(256), 0, st, d_hist, (u64)max_bin + 1, d_full); if (hipGetLastError() != hipSuccess) arc = LRGE_ERR_DEVICE; }
            cg.disarm();
            rc = comm_agree(c, arc, st); if (rc) return rc;
            rc = comm_allreduce_sum(c, d_full, (size_t)max_bin + 1, 8, st); if (rc) return rc;
            std::vector<u64> full((size_t)max_bin + 1);
            HIPCHK(ctx, hipMemcpyAsync(full.data(), d_full, full.size() * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            cum = 0;
            for (u32 b = 0; b <= max_bin; ++b) { cum += full[b]; if (cum > kth) { v = b; break; } }
        }
        thres = (int)v + 1;
    }
    if (thres < P.min_mid_occ) thres = P.min_mid_occ;
    if (P.max_mid_occ > P.min_mid_occ && thres > P.max_mid_occ) thres = P.max_mid_occ;
    // (tests: DEBUG_TS_MID_OCC forces the threshold -- on every rank alike -- so that tiny clean data has too-frequent keys at all)
    const u32 mid_occ = (u32)ctx->opt_u64("DEBUG_TS_MID_OCC", (u64)thres);
    // ---- C5, A2, C6, A3: the too-frequent keys go to everybody ----
    // the list starts at 2^16 keys (mid_occ_frac = 2 x 10^-4 of the keys an owner holds: thousands at H. sapiens scale); an owner with
    // more -- a repeat-rich set -- takes a list of exactly that many and lists again (round 5: was a fixed 32 MB block and a hard
    // limit of 2^22 keys)
    u32 cap_list = (u32)std::min<u64>(std::max<u64>(n_r, 1), ctx->opt_u64("DEBUG_TS_LIST_CAP", 1u << 16));
    u64 *d_list = nullptr; u32 *d_nf = nullptr;
    std::vector<u64> mine2(2, 0), all2((size_t)2 * W, 0);
    cg.expect(CollectiveGuard::ALLGATHER_U64, 2, 1);
    auto local3 = [&]() -> int {
        if (shard_fail_at(ctx, 14)) return LRGE_ERR_DEVICE;
        d_nf = sc.get<u32>(TS_MAX_WORLD + 1);
        if (!d_nf) return LRGE_ERR_DEVICE;
        for (int pass = 0; pass < 2; ++pass) {
            d_list = sc.get<u64>(cap_list);
            if (!d_list) return LRGE_ERR_DEVICE;
            HIPCHK(ctx, hipMemsetAsync(d_nf, 0, 4, st));
            if (n_r) { hipLaunchKernelGGL(k_ts_frequent, dim3((u32)div_up(n_r, 256)), dim3(256), 0, st, rk, starts, d_nr, gcnt, mid_occ, d_list, cap_list, d_nf); KCHK(ctx); }
            u32 nf = 0;
            HIPCHK(ctx, hipMemcpyAsync(&nf, d_nf, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            mine2[0] = nf;
            if (nf <= cap_list) return LRGE_OK;
            sc.drop(d_list); d_list = nullptr;
            cap_list = nf;                                  // (the count is exact: the second pass fits)
        }
        LRGE_SET_ERR(ctx, "target-sharded index build: the list of too-frequent keys did not settle");
        return LRGE_ERR_DEVICE;
    };
    rc = local3();
    const bool failed3 = rc != LRGE_OK;
    mine2[1] = failed3 ? 1 : 0;
    cg.disarm();
    rc = comm_allgather_host(c, mine2.data(), 16, all2.data(), st); if (rc) return rc;
    u64 max_nf = 0;
    for (int r = 0; r < W; ++r) {
        if (all2[(size_t)2 * r + 1]) { if (!failed3) LRGE_SET_ERR(ctx, "target-sharded index build: rank %d failed", r); return LRGE_ERR_DEVICE; }
        max_nf = std::max(max_nf, all2[(size_t)2 * r]);
    }
    // From here on nothing a rank does is known to the others unless it says so: the lists' all-gather, the marking kernels and the
    // last synchronise can fail on one rank alone, and a rank that returned LRGE_OK goes on to the all-reduce that closes the step
    // while the failed one does not.  One more agreement (A3) closes the build: every rank leaves with the same verdict.
    cg.expect(CollectiveGuard::AGREE);
    auto tail = [&]() -> int {
        if (max_nf) {
            // everything that can fail on this rank alone between C5 and A2 -- the re-taken list, the room for everybody's lists, the
            // counts' copy -- lands in `arc`: a rank ALWAYS enters A2 before A3 when max_nf != 0 (a rank that returned from
            // here went straight to A3 while its peers sat in A2, and over RCCL / host callbacks they would have waited for ever)
            int arc = LRGE_OK;
            // (the all-gather reads max_nf words of EVERY rank's list: a list that is shorter is re-taken at that size)
            const bool f17 = shard_fail_at(ctx, 17);        // (tests: the re-taken list refused -- or, on the rank whose list is the longest, the room for everybody's)
            if (!(cap_list < max_nf) && f17) arc = LRGE_ERR_DEVICE;
            if (cap_list < max_nf) {
                u64 *bigger = f17 ? nullptr : sc.get<u64>(max_nf);
                if (!bigger) arc = LRGE_ERR_DEVICE;
                else {
                    if (mine2[0] && hipMemcpyAsync(bigger, d_list, mine2[0] * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipGetLastError(); LRGE_SET_ERR(ctx, "target-sharded index build: copying the list of too-frequent keys failed"); arc = LRGE_ERR_DEVICE; }
                    d_list = bigger;
                }
            }
            u64 *d_all = arc == LRGE_OK ? sc.get<u64>(max_nf * (u64)W) : nullptr;
            std::vector<u32> nof(TS_MAX_WORLD, 0);
            for (int r = 0; r < W; ++r) nof[(size_t)r] = (u32)all2[(size_t)2 * r];
            if (!d_all) arc = LRGE_ERR_DEVICE;
            if (d_all && hipMemcpyAsync(d_nf + 1, nof.data(), TS_MAX_WORLD * 4, hipMemcpyHostToDevice, st) != hipSuccess) arc = LRGE_ERR_DEVICE;
            if (d_all && hipStreamSynchronize(st) != hipSuccess) arc = LRGE_ERR_DEVICE;
            if (shard_fail_at(ctx, 15)) arc = LRGE_ERR_DEVICE;
            int r2 = comm_agree(c, arc, st); if (r2) return r2;                                   // A2
            r2 = comm_allgather(c, d_list, max_nf * 8, d_all, st); if (r2) return r2;             // C6
            if (shard_fail_at(ctx, 16)) return LRGE_ERR_DEVICE;
            for (const TsTable &t : tabs) {
                hipLaunchKernelGGL(k_ts_mark, dim3((u32)div_up(max_nf, 256), (u32)W), dim3(256), 0, st, t.ht, t.cap, fix, d_all, d_nf + 1, (u32)W, (u32)max_nf, mid_occ);
                KCHK(ctx);
            }
        } else if (shard_fail_at(ctx, 16)) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    };
    rc = tail();
    cg.disarm();
    rc = comm_agree(c, rc, st); if (rc) return rc;                                                // A3
    mark("frequent keys + marking");
    ix->mid_occ = (int)mid_occ; ix->n_keys = g_distinct; ix->n_mz = g_mz;
    for (lrge_hip_index *p : ix->parts) p->mid_occ = (int)mid_occ;
    return LRGE_OK;
}

extern "C" int lrge_hip_index_build_tsharded(lrge_hip_ctx *ctx, const lrge_hip_seqset *target_shard, int preset, lrge_hip_comm *comm, lrge_hip_index **out) {
    if (!ctx || !target_shard || !comm || !out) return LRGE_ERR_INVALID;
    *out = nullptr;
    // (argument errors are rank-local by nature -- every rank passes the same job -- so they return before any collective)
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    if (target_shard->ctx != ctx || comm->ctx != ctx) { LRGE_SET_ERR(ctx, "index_build_tsharded: set / communicator belong to another context"); return LRGE_ERR_INVALID; }
    if (comm->world > TS_MAX_WORLD) { LRGE_SET_ERR(ctx, "index_build_tsharded: at most %d ranks", TS_MAX_WORLD); return LRGE_ERR_INVALID; }
    // from here on a failure is owed to the build's first collective: the (world + 1)-word counts all-gather
    CollectiveGuard cg{comm, ctx->stream};
    cg.expect(CollectiveGuard::ALLGATHER_U64, (size_t)comm->world + 1, (size_t)comm->world);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (shard_fail_at(ctx, 10)) return LRGE_ERR_DEVICE;
    lrge_hip_index *ix = nullptr;
    ctx->ts_build = true;                         // (a partitioned local index leaves its occurrence statistics to ts_global_stats)
    int rc = lrge_hip_index_build(ctx, target_shard, preset, &ix);
    ctx->ts_build = false;
    if (rc) return rc;
    IndexGuard g(ix);
    float ms_keep[LRGE_T_N]; u64 cn_keep[LRGE_C_N];
    memcpy(ms_keep, ctx->ms, sizeof ms_keep); memcpy(cn_keep, ctx->counters, sizeof cn_keep);
    hipEvent_t e0 = ctx->get_event(), e1 = ctx->get_event();
    (void)hipEventRecord(e0, ctx->stream);
    rc = ts_global_stats(ctx, ix, comm, cg);
    (void)hipEventRecord(e1, ctx->stream);
    if (rc) { ctx->event_pool.push_back(e0); ctx->event_pool.push_back(e1); return rc; }
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    ctx->event_pool.push_back(e0); ctx->event_pool.push_back(e1);
    memcpy(ctx->ms, ms_keep, sizeof ms_keep); memcpy(ctx->counters, cn_keep, sizeof cn_keep);
    ctx->ms[LRGE_T_INDEX_RESTRICT] += ms; ctx->ms[LRGE_T_TOTAL] += ms;
    *out = g.release();
    return LRGE_OK;
}
