// host_overlap_batch.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): the overlap core, second half: one batch of queries -- K4 expansion (with the dead-pair filter), anchor sort plan and sort, K5 groups, K6 chain, K7 count -- and the results to the host.
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
            {   // what the planner of the next batches / calls assumes (OverlapRun::plan): the largest share seen lately
                const double r_ = (double)A / (double)A_all;
                ctx->kept_ratio = ctx->kept_seen ? std::max(r_, 0.9 * ctx->kept_ratio) : r_;
                ctx->kept_seen = true;
            }
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
                    la.slow_rate = (u32)ctx->opt_u64("LPG_SLOW_RATE", 4);
                    la.slow_entry_every = (u32)ctx->opt_u64("LPG_SLOW_ENTRY_EVERY", 32);
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

