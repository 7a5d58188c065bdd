// host_index_build.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): index_build_one: sketch -> (restriction + global statistics) -> radix sort -> run heads -> ordered table -> mid_occ, for one index (a whole set, one part of a partitioned index, one rank's restricted index).
// How many more hash bits than the low byte the SEGMENT of a segment-packed entry implies (index_sort_segpacked): `over`, what the word
// is too narrow by (full-size C5 in 3 parts has read ids of 20 bits: 2 too many); beyond that, e is chosen so that what is LEFT of the
// hash, nr = 2k - 8 - e bits, takes as few 8-bit LSD passes as possible (round 5: k = 19 -> e = 6, nr = 24: passes A + A2 + 3 instead of
// A + A2 + 4; k = 15 -> e = 0, 22 bits in 3 passes either way).  DEBUG_SEG_EXTRA forces an e on small sets; SEG_PACK_EXTRA_MAX caps it
// (0: never an A2 pass).  ~0u: no e will do.
static u32 seg_pack_extra(lrge_hip_ctx *ctx, const Preset &P, u32 yb_p) {
    const u32 over = 2 * (u32)P.k - 8 + yb_p > 64 ? 2 * (u32)P.k - 8 + yb_p - 64 : 0;
    const u32 e_cap = std::min<u32>(7, (u32)ctx->opt_u64("SEG_PACK_EXTRA_MAX", 7));
    u32 extra = over;
    if (ctx->opt("DEBUG_SEG_EXTRA")) extra = std::max<u32>(over, (u32)ctx->opt_u64("DEBUG_SEG_EXTRA", 0));
    else {
        auto n_pass = [&](u32 e) { return 1u + (e ? 1u : 0u) + (2 * (u32)P.k - 8 - e + 7) / 8; };
        for (u32 e = over + 1; e <= e_cap && 2 * (u32)P.k > 16 + e; ++e) if (n_pass(e) < n_pass(extra)) extra = e;
    }
    const bool extra_ok = extra == 0 || (extra <= e_cap && 2 * (u32)P.k > 16 + extra);
    return extra_ok ? extra : ~0u;
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
    bool segw = false;          // the sketch wrote SEGW entries (below)
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
        // SEGW entries (round 5): a build that will take the segment-packed sort below -- pairs too wide for one word, a plain build, a set
        // large enough -- has its sketch write [word, 16-bit digit pair] instead of (hash, y): 12 bytes per entry instead of 16 through
        // the slots, the compaction and the sort's first two passes (k_sketch.h PK == 2; k_prims.h index_sort_segw).  Decided before
        // the count is known: from the bases, at the density the part planner assumes.  option NO_SEGW: pairs, as rounds 3-4.
        {
            const u32 yb_p = pk_rid + pk_pos1;
            const u32 extra = seg_pack_extra(ctx, P, yb_p);
            const double dens = P.hpc ? 0.24 : 0.32;
            segw = !pk && !ro && extra != ~0u && 2 * P.k >= 24 && !ctx->opt("NO_SEG_PACK") && !ctx->opt("NO_SEGW") && 2 * (u32)P.k - 16 + yb_p <= 64 &&
                   (double)targets->total_bases * dens >= (double)ctx->opt_u64("SEG_PACK_MIN", 1ULL << 22);
        }
        rc = sketch_device(ctx, sc, targets, preset, true, &so, (pk || segw) ? pk_pos1 : 0, segw ? pk_rid + pk_pos1 : pk_ybits, nullptr, keep_slots, segw, /*wave_ok=*/true);
        if (rc) return rc;
        if (so.mz_off) sc.drop(so.mz_off);
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
    bool seg_packed = false; u32 kshift_t = pk_ybits, seg_e = 0; u32 *d_seg_start = nullptr; std::vector<u32> h_seg_start;
    {
        StageTimer t(ctx, LRGE_T_INDEX_SORT);
        ALLOC_OR_FAIL(k1, sc, u64, M + 1);
        if (pk && so.wave_x) {
            // packed entries out of the wave-dense sketch: the first pass reads the wavefronts' slots (one 32-byte descriptor per tile)
            if (M) {
                ALLOC_OR_FAIL(k0, sc, u64, M + 1);
                rc = radix_sort_keys_first_pass_from_slots(ctx, sc, so.wave_x, so.wave_offs, so.n_waves, so.wave_cap, k1, M, (int)pk_ybits, 2 * P.k, /*reverse_digits=*/true);
                if (rc) return rc;
                sc.drop(so.wave_x); sc.drop(so.wave_cnt); sc.drop(so.wave_offs);
                u64 *rk = k1;
                rc = radix_sort_keys(ctx, sc, k1, k0, M, (int)pk_ybits, 2 * P.k, &rk, /*reverse_digits=*/true, 1, -1);
                if (rc) return rc;
                skey = rk; spos = rk;
                sc.drop(rk == k1 ? k0 : k1);
            } else { sc.drop(so.wave_x); sc.drop(so.wave_cnt); sc.drop(so.wave_offs); skey = spos = k1; }
        } else if (pk && so.slots) {
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
            // (v1, the second buffer of the pair layout, is taken by the branches that sort pairs only: on the SEGW path -- the default
            // at H. sapiens scale -- it would be 27 GB of transient peak beside the 24 B per entry that path needs)
            u64 *v1 = nullptr;
            // the pair layout, segment-packed (k_prims.h: index_sort_segpacked): behind the first digit the low hash byte is implied
            // and the rest of the entry fits one word -- fewer bytes through the remaining passes, 8 bytes per entry resident
            const u32 yb_p = pk_rid + pk_pos1;
            // `over` more hash bits must be implied by the segment for the word to fit (full-size C5 in 3 parts has read ids of 20 bits:
            // 2 too many); beyond that, e is chosen so that what is LEFT of the hash, nr = 2k - 8 - e bits, takes as few 8-bit LSD
            // passes as possible (round 5: k = 19 -> e = 6, nr = 24: passes A + A2 + 3 instead of A + A2 + 4; k = 15 -> e = 0, 22 bits
            // in 3 passes either way).  DEBUG_SEG_EXTRA forces an e on small sets; SEG_PACK_EXTRA_MAX caps it (0: never an A2 pass).
            const u32 extra_ = seg_pack_extra(ctx, P, yb_p);
            const bool extra_ok = extra_ != ~0u;
            const u32 extra = extra_ok ? extra_ : 0;
            if (so.segw && M == 0) {
                // (nothing to sort: an empty pair stream is an empty SEGW stream; the wave-dense sketch left no dense arrays at all)
                if (so.wave_x) { sc.drop(so.wave_x); sc.drop(so.wave_d); sc.drop(so.wave_cnt); sc.drop(so.wave_offs); skey = spos = k1; }
            } else if (so.segw) {
                // (the sketch already wrote what pass A wants: decided above, whatever M turned out to be)
                const bool wave = so.wave_x != nullptr;          // (the wave-dense sketch: 16-bit DIG members, pass A reads the wavefronts' slots)
                u32 *d1 = wave ? (u32 *)sc.get<wdig_t>(M + 2) : sc.get<u32>(M + 1);
                if (!d1) return LRGE_ERR_DEVICE;
                u64 *rk = nullptr, *spare_ = nullptr;
                WaveSrc ws{so.wave_x, so.wave_d, so.wave_cnt, so.wave_offs, so.n_waves, so.wave_cap};
                rc = index_sort_segw(ctx, sc, so.x, (u32 *)so.y, k1, d1, M, 2 * P.k, yb_p, pk_pos1, extra, &rk, &d_seg_start, wave ? &ws : nullptr, &spare_);
                if (rc) return rc;
                h_seg_start.assign((256u << extra) + 1, 0u);
                HIPCHK(ctx, ctx->d2h(h_seg_start.data(), d_seg_start, h_seg_start.size() * 4, ctx->stream));
                seg_packed = true; kshift_t = yb_p; seg_e = extra;
                skey = rk; spos = rk;
                sc.drop(spare_); if (so.y) sc.drop((u32 *)so.y); sc.drop(d1);
            } else if (pass_from == 0 && extra_ok && 2 * P.k > 16 && !ctx->opt("NO_SEG_PACK") && M >= ctx->opt_u64("SEG_PACK_MIN", 1ULL << 22)) {
                v1 = sc.get<u64>(M + 1); if (!v1) return LRGE_ERR_DEVICE;
                u64 *rk = nullptr;
                rc = index_sort_segpacked(ctx, sc, so.x, so.y, k1, v1, M, 2 * P.k, yb_p, pk_pos1, extra, &rk, &d_seg_start);
                if (rc) return rc;
                h_seg_start.assign((256u << extra) + 1, 0u);              // (the dump wants them: they arrive with the table build's first round trip)
                HIPCHK(ctx, ctx->d2h(h_seg_start.data(), d_seg_start, h_seg_start.size() * 4, ctx->stream));
                seg_packed = true; kshift_t = yb_p; seg_e = extra;
                skey = rk; spos = rk;
                sc.drop(rk == so.x ? so.y : so.x); sc.drop(k1); sc.drop(v1);
            } else {
            v1 = sc.get<u64>(M + 1); if (!v1) return LRGE_ERR_DEVICE;
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
            rc = compact_heads(ctx, sc, skey, M, kshift_t, &d_runstart, &n_runs, d_seg_start, seg_packed ? 256u << seg_e : 0u);    // runs of equal hash
            if (rc) return rc;
        }
#ifndef HT_CAP_NUM
#define HT_CAP_NUM 2       // home slots per distinct key = HT_CAP_NUM / HT_CAP_DEN
#define HT_CAP_DEN 1
#endif
        // a part of a partitioned index (a target set of tens of gigabases) gets 1.5 instead of 2 slots per key: the tables of all
        // parts are resident together.  (1.25 until round 5; swept at full-size C5 ava-pb, profiles/r05_htcap_ab.txt: k_lookup 42.6 ms
        // per step at 1.25, 34.4 at 1.5, 31.2 at 2.0 -- where the table passes give 7 ms back: 1.5 is the sum's minimum, for 1.6 GB)
        u64 cap = targets->is_view ? (u64)n_runs * 3 / 2 : (u64)n_runs * HT_CAP_NUM / HT_CAP_DEN;
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
                hipLaunchKernelGGL(k_place_reduce, dim3(n_tiles), dim3(PLACE_THREADS), 0, ctx->stream, skey, d_runstart, n_runs, cap, bmax, kshift_t, ht_fix, SegStarts{d_seg_start, seg_e, 2 * (u32)P.k});
                KCHK(ctx);
                hipLaunchKernelGGL(k_place_scan, dim3(1), dim3(1024), 0, ctx->stream, bmax, n_tiles);
                KCHK(ctx);
                hipLaunchKernelGGL(k_place_apply, dim3(std::min<u32>(n_tiles, (u32)ctx->n_cu * 8)), dim3(PLACE_THREADS), 0, ctx->stream,
                                   skey, d_runstart, n_runs, M, cap, n_slots, bmax, ht, d_occ, max_bin, d_occ + max_bin + 1, kshift_t, ht_fix,
                                   fused_fill ? bmax + n_tiles : (u32 *)nullptr, pk_t ? (const u64 *)nullptr : (const u64 *)spos, pk_t ? pk_pos1 : 0u,
                                   ctx->opt("NO_INLINE_SINGLETONS") ? 0u : 1u, SegStarts{d_seg_start, seg_e, 2 * (u32)P.k});
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
    if (seg_packed) { ix->h_seg_start = h_seg_start; ix->seg_e = seg_e; sc.drop(d_seg_start); }     // (the device copy served the table build; the dump needs the host copy)
    t_total.stop();
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->resolve_timers();
    pool_report(ctx, "index_build_one");
    *out = ix_guard.release();
    return LRGE_OK;
}

