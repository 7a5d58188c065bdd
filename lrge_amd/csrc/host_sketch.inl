// host_sketch.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): the sketch driver: one-pass / two-pass / gated launches of k_sketch.h, and the presketch of a streamed set on the side stream.
// ------------------------------------------------------------------------------------------
// sketch driver
// ------------------------------------------------------------------------------------------
struct SketchOut {
    u64 *x = nullptr, *y = nullptr;   // pool memory (owned by the caller's Scratch)
    u32 *mz_off = nullptr;            // [n+1] per-read offsets
    u64 n = 0;
    // keep_slots (packed index entries, one-pass form): x stays null -- the entries still sit in the per-chunk slots of k_sketch_direct
    // (chunk c at slots[c * SK_CAP ...), offs[] = exclusive scan of the per-chunk counts) and the index sort's first pass reads them there
    u64 *slots = nullptr; u32 *offs = nullptr; u32 n_chunks = 0;
    bool segw = false;                // y holds a u32 ARRAY: SEGW entries (k_sketch.h, sketch_write_chunk PK == 2): x = word, y[i] = the two sort digits
    // wave-dense form (k_sketch.h: k_sketch_wave; SEGW entries only): x / y stay null -- wavefront v's entries sit, densely and in emission order,
    // at wave_x / wave_d [v * wave_cap ...), wave_cnt[v] of them; the index sort's first pass reads them there (k_prims.h: index_sort_segw, WaveSrc)
    u64 *wave_x = nullptr; wdig_t *wave_d = nullptr; u32 *wave_cnt = nullptr, *wave_offs = nullptr; u32 n_waves = 0, wave_cap = 0;      // wave_offs: exclusive scan of wave_cnt
};

// pk_ybits != 0 (index only): packed 8-byte entries in o->x, o->y stays null (k_sketch.h PK)
template <int K, int W, bool HPC>
static int sketch_launch(lrge_hip_ctx *ctx, Scratch &sc, const lrge_hip_seqset *s, bool index_keys, SketchOut *o, u32 pk_pos1, u32 pk_ybits,
                         std::vector<u32> *h_mzoff, bool gated = false, bool keep_slots = false, bool segw = false, bool wave_ok = false) {
    // gated: the caller has NOT waited for the set's upload (seqset_ready): this function does, as late as it can -- chunk range by
    // chunk range behind the upload's gates where the form allows it
    if (s->n_chunks >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "read set too large for one sketch launch"); return LRGE_ERR_TOO_MANY; }
    u32 n_chunks = (u32)s->n_chunks;
    // Option SKETCH_TILE_FORM: k_sketch_tile (k_sketch_tile.h: a lane per step, a workgroup per 16 chunks; + k_sketch_redo for the HPC tiles
    // it marks) in place of k_sketch_direct (a lane per chunk) in the one-pass forms, into the same per-chunk slots.  Exact, and NOT the
    // default: measured at full-size C5 it is slower with HPC (index sketch 286 against 198 ms) and equal without.
    const bool tile_form = ctx->opt("SKETCH_TILE_FORM") != nullptr && !segw;      // (the tile form writes pairs or packed words only)
    // the upload job whose gates cover this set's words (a view: its root's), and how far the sketch may go behind gate j:
    // every chunk that lies wholly inside the words that have arrived -- with HPC only the chunks of reads that have arrived
    // WHOLLY (a homopolymer-compressed step may run past its chunk, to the end of the read at most)
    std::shared_ptr<UploadJob> gjob;
    if (gated) {
        const lrge_hip_seqset *root = s->is_view ? s->view_root : s;
        gjob = s->is_view ? s->view_job : s->job;
        if (!gjob || !root || root->job != gjob || gjob->gate_ev.empty()) { gjob.reset(); }
    }
    auto chunks_behind = [&](u64 w1) -> u32 {        // w1: absolute word offset the gate covers up to
        if (w1 >= s->h_woff[s->n]) return n_chunks;
        if (w1 <= s->h_woff[0]) return 0;
        const u32 r = (u32)(std::upper_bound(s->h_woff.begin(), s->h_woff.end(), w1) - s->h_woff.begin()) - 1;
        if (HPC) return s->h_cs[r];
        u64 avail = w1 - s->h_woff[r];                            // words of read r that have arrived: 4 per 128-base chunk
        if (tile_form && avail) --avail;                          // (the tile form looks w steps past a chunk's end: one word more)
        return s->h_cs[r] + (u32)std::min<u64>(avail / (SK_CHUNK / 32), (u64)(s->h_cs[r + 1] - s->h_cs[r]));
    };
    // the first gate behind which chunks [.., c1) of this set may be sketched (waited for on the host until its transfer has been
    // queued, then on the device); false: the upload failed
    size_t g_next = 0;
    auto wait_chunks = [&](u32 c1) -> int {
        if (!gjob) return LRGE_OK;
        const size_t ng = gjob->gate_w1.size();
        size_t j = g_next;
        while (j + 1 < ng && chunks_behind(gjob->gate_w1[j]) < c1) ++j;
        if (!gjob->wait_gate((int)j)) return LRGE_ERR_DEVICE;       // (the job failed: seqset_ready reports it)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, gjob->gate_ev[j], 0));
        g_next = j;
        return LRGE_OK;
    };
    const u32 *d_cs = s->d_cs;           // chunk map, uploaded with the set
    const bool pk = index_keys && pk_ybits && !segw;
    const u64 ebytes = pk ? 8 : segw ? 12 : 16;          // bytes per entry in the slots and in the output
    auto get_y = [&](size_t n_) -> u64 * { return segw ? (u64 *)sc.get<u32>(n_) : sc.get<u64>(n_); };      // the second member's array
    ALLOC_OR_FAIL(d_cnt, sc, u32, (size_t)n_chunks + 1);
    ALLOC_OR_FAIL(d_total, sc, u32, 2);  // [1] = overflow flag of the one-pass form
    ALLOC_OR_FAIL(d_mzoff, sc, u32, (size_t)s->n + 1);
    ChunkMap cm{d_cs, s->n};
    const dim3 sgrid((u32)div_up(n_chunks, SK_THREADS));
    u32 tot_ovf[2] = {0, 0};
    // ---- wave-dense form (round 6): index entries (SEGW, or packed words for the sort that reads slots) written densely per wavefront, no
    // compaction; see k_sketch_wave ----
    // (option NO_WAVE_SKETCH: the slot-per-chunk forms below, rounds 2-5; WAVE_CAP: entries of room per wavefront, a multiple of 64, <= RS_TILE)
    if (wave_ok && (segw || (pk && keep_slots)) && n_chunks && !tile_form && !ctx->opt("DEBUG_SK_CAP") && !ctx->opt("NO_WAVE_SKETCH") && !ctx->opt("SKETCH_TWO_PASS") && !ctx->opt("DEBUG_SK_RANGE_CHUNKS")) {
        const u32 n_waves = (u32)div_up(n_chunks, 64);
        u32 capw = (u32)ctx->opt_u64("WAVE_CAP", HPC ? 2560 : 3328);
        capw = std::max<u32>(64, capw / 64 * 64);
        u64 *wx = sc.get<u64>((size_t)n_waves * capw + 8);
        wdig_t *wd = (wx && segw) ? sc.get<wdig_t>((size_t)n_waves * capw + 8) : nullptr;      // (packed 8-byte entries have no digit member)
        u32 *wcnt = (wx && (wd || !segw)) ? sc.get<u32>((size_t)n_waves + 1) : nullptr, *woffs = wcnt ? sc.get<u32>((size_t)n_waves + 1) : nullptr;
        if (wx && (wd || !segw) && wcnt && woffs) {
            HIPCHK(ctx, hipMemsetAsync(d_total, 0, 8, ctx->stream));
            auto launch_wave = [&](u32 c0, u32 c1) {           // chunks [c0, c1), c0 a multiple of 64
                if (c1 <= c0) return;
                StageTimer tk(ctx, LRGE_T_K_SKETCH);
                ctx->counters[LRGE_C_SKETCH_WAVE_LAUNCHES] += 1;
                if (segw) hipLaunchKernelGGL((k_sketch_wave<K, W, HPC, 2>), dim3((u32)div_up(c1 - c0, SK_THREADS)), dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,
                                             s->d_len, cm, c1, wcnt, d_total + 1, wx, wd, pk_pos1, pk_ybits, capw, c0);
                else hipLaunchKernelGGL((k_sketch_wave<K, W, HPC, 1>), dim3((u32)div_up(c1 - c0, SK_THREADS)), dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,
                                        s->d_len, cm, c1, wcnt, d_total + 1, wx, (wdig_t *)nullptr, pk_pos1, pk_ybits, capw, c0);
            };
            u32 c_prev = 0;
            if (gated && gjob) {
                // (the set's upload is still in flight: the wavefronts whose 64 chunks lie wholly inside the words of upload chunk j run behind gate j)
                const size_t ng = gjob->gate_w1.size();
                for (size_t j = 0; j < ng && c_prev < n_chunks; ++j) {
                    u32 c_end = chunks_behind(gjob->gate_w1[j]);
                    if (c_end < n_chunks) c_end &= ~63u;
                    if (c_end <= c_prev) continue;
                    if (!gjob->wait_gate((int)j)) break;                         // (the job failed: seqset_ready below reports it)
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, gjob->gate_ev[j], 0));
                    launch_wave(c_prev, c_end);
                    KCHK(ctx);
                    c_prev = c_end;
                }
            }
            if (gated) { int rr = seqset_ready(ctx, s); if (rr) return rr; gated = false; }
            launch_wave(c_prev, n_chunks);
            KCHK(ctx);
            int rc = scan_exclusive_u32(ctx, sc, wcnt, woffs, n_waves, d_total);
            if (rc) return rc;
            HIPCHK(ctx, ctx->d2h(tot_ovf, d_total, 8, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            if (!tot_ovf[1]) {
                if (h_mzoff) h_mzoff->clear();
                sc.drop(d_cnt); sc.drop(d_total); sc.drop(d_mzoff);
                o->x = nullptr; o->y = nullptr; o->mz_off = nullptr; o->n = tot_ovf[0]; o->segw = segw;
                o->wave_x = wx; o->wave_d = wd; o->wave_cnt = wcnt; o->wave_offs = woffs; o->n_waves = n_waves; o->wave_cap = capw;
                return LRGE_OK;
            }
            tot_ovf[0] = tot_ovf[1] = 0;          // a wavefront found more than its slot holds: the slot-per-chunk forms
            sc.drop(wx); if (wd) sc.drop(wd); sc.drop(wcnt); sc.drop(woffs);
        } else {
            if (wx) sc.drop(wx); if (wd) sc.drop(wd); if (wcnt) sc.drop(wcnt); if (woffs) sc.drop(woffs);
            (void)hipGetLastError(); ctx->err.clear();
        }
    }
    // One pass (k_sketch_direct into per-chunk slots, then k_sketch_compact) when the slots fit comfortably; the
    // two-pass form (count, scan, write) otherwise, when a chunk overflows its slot, or on request.
    const u64 slot_bytes = (u64)n_chunks * SK_CAP * ebytes;
    size_t mfree = (size_t)64 << 30, mtot = 0;
    if (slot_bytes > ((u64)4 << 30)) (void)hipMemGetInfo(&mfree, &mtot);       // (small sets: no need to ask)
    bool one_pass = n_chunks && !ctx->opt("SKETCH_TWO_PASS") && slot_bytes < ((u64)mfree + ctx->pool.idle()) / 4 && !ctx->opt("DEBUG_SK_RANGE_CHUNKS");   // (tests: the ranged form)
    const char *cap_env = ctx->opt("DEBUG_SK_CAP");                      // tests: force the overflow fallback
    const u32 sk_cap = cap_env ? (u32)std::min<u64>(strtoull(cap_env, nullptr, 10), SK_CAP) : (u32)SK_CAP;
    u64 *tx = nullptr, *ty = nullptr;
    if (one_pass) {
        tx = sc.get<u64>((size_t)n_chunks * SK_CAP);
        ty = pk ? nullptr : get_y((size_t)n_chunks * SK_CAP);
        if (!tx || (!pk && !ty)) { if (tx) sc.drop(tx); if (ty) sc.drop(ty); tx = ty = nullptr; one_pass = false; (void)hipGetLastError(); }
    }
    // chunks [c0, c1) into the slots sx / sy (indexed by chunk number), counts to d_cnt, overflow flag at d_total + 1
    auto launch_slots = [&](u32 c0, u32 c1, u64 *sx, u64 *sy, u32 capv) {
        if (c1 <= c0) return;
        const dim3 gl((u32)div_up(c1 - c0, SK_THREADS)), gt((u32)div_up(c1 - c0, ST_G));
#define LRGE_SK_LAUNCH(IK, PKF, P1, YB)                                                                                                             \
        do {                                                                                                                                        \
            if (tile_form) {                                                                                                                        \
                hipLaunchKernelGGL((k_sketch_tile<K, W, HPC, IK, PKF>), gt, dim3(ST_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,      \
                                   s->d_len, cm, c1, d_cnt, d_total + 1, sx, sy, P1, YB, capv, c0);                                                  \
                if constexpr (HPC)                                                                                                                  \
                    hipLaunchKernelGGL((k_sketch_redo<K, W, IK, PKF>), gl, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,       \
                                       s->d_len, cm, c1, d_cnt, d_total + 1, sx, sy, P1, YB, capv, c0);                                              \
            } else                                                                                                                                  \
                hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, IK, PKF>), gl, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,    \
                                   s->d_len, cm, c1, d_cnt, d_total + 1, sx, sy, P1, YB, capv, c0);                                                  \
        } while (0)
        StageTimer tk(ctx, LRGE_T_K_SKETCH);             // (timer level 2: the bench's roofline candidates)
        if (!tile_form) ctx->counters[LRGE_C_SKETCH_LAUNCHES] += 1;
        if (pk) LRGE_SK_LAUNCH(true, true, pk_pos1, pk_ybits);
        else if (segw)          // (the lane form only: tile_form is off for SEGW entries)
            hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, 2>), gl, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff,
                               s->d_len, cm, c1, d_cnt, d_total + 1, sx, sy, pk_pos1, pk_ybits, capv, c0);
        else if (index_keys) LRGE_SK_LAUNCH(true, false, 0u, 0u);
        else LRGE_SK_LAUNCH(false, false, 0u, 0u);
#undef LRGE_SK_LAUNCH
    };
    // ---- ranged one-pass form: the slots of the whole set do not fit, those of a range of its chunks do ----
    // Range after range: k_sketch_direct into the SAME slots, scan of the range's counts, compaction behind what the ranges before
    // left.  The output is sized by an estimate (the count is only known at the end); a set that beats the estimate, or a chunk
    // that overflows its slot, starts over in the two-pass form below.  Full-size C5: index sketch of a 10-Gbase part and the
    // streamed views of the inverse strategy (two passes: 8.3 ps per base; one pass + compaction: 7.3).
    if (n_chunks && !one_pass && !ctx->opt("SKETCH_TWO_PASS") && !ctx->opt("NO_RANGED_SKETCH") && !cap_env) {
        const u64 per_chunk = (u64)SK_CAP * ebytes;
        const u64 avail = (u64)mfree + ctx->pool.idle();
        u64 R = std::min<u64>(avail / 8, (u64)24 << 30) / per_chunk / 256 * 256;
        R = ctx->opt_u64("DEBUG_SK_RANGE_CHUNKS", R);
        const u64 est = std::min<u64>((u64)s->total_bases + 1, (u64)((double)s->total_bases * 0.40) + 65536);
        if (R >= 256 && R < n_chunks && est < (1ULL << 32) && est * ebytes < avail / 2) {
            if (gated && !gjob) { int rr = seqset_ready(ctx, s); if (rr) return rr; gated = false; }
            u64 *rx = sc.get<u64>((size_t)R * SK_CAP), *ry = pk ? nullptr : get_y((size_t)R * SK_CAP);
            u64 *dx = sc.get<u64>((size_t)est + 1), *dy = pk ? nullptr : get_y((size_t)est + 1);
            u32 *d_run = sc.get<u32>(2);                  // [0] output offset behind the ranges done, [1] the current range's count
            if (rx && (pk || ry) && dx && (pk || dy) && d_run) {
                HIPCHK(ctx, hipMemsetAsync(d_total, 0, 8, ctx->stream));
                HIPCHK(ctx, hipMemsetAsync(d_run, 0, 8, ctx->stream));
                for (u64 c0 = 0; c0 < n_chunks; c0 += R) {
                    const u32 c1 = (u32)std::min<u64>(c0 + R, n_chunks), len = c1 - (u32)c0;
                    if (gated && wait_chunks(c1) != LRGE_OK) { int rr = seqset_ready(ctx, s); return rr ? rr : LRGE_ERR_DEVICE; }   // this range's reads have arrived; the later ones may still travel
                    u64 *sx = rx - c0 * SK_CAP, *sy = !ry ? nullptr : segw ? (u64 *)((u32 *)ry - c0 * SK_CAP) : ry - c0 * SK_CAP;      // (the kernels index slots by chunk number)
                    launch_slots((u32)c0, c1, sx, sy, (u32)SK_CAP);
                    KCHK(ctx);
                    int rc = scan_exclusive_u32(ctx, sc, d_cnt + c0, d_cnt + c0, len, d_run + 1);
                    if (rc) return rc;
                    hipLaunchKernelGGL(k_add_base_u32, dim3((u32)div_up(len, 256)), dim3(256), 0, ctx->stream, d_cnt + c0, len, d_run);
                    hipLaunchKernelGGL(k_bump_u32, dim3(1), dim3(1), 0, ctx->stream, d_run, d_run + 1, d_total + 1);
                    KCHK(ctx);
                    const dim3 cgrid((u32)div_up(div_up(len, 64), 4));
                    if (pk) hipLaunchKernelGGL(k_sketch_compact<false>, cgrid, dim3(256), 0, ctx->stream, sx, sy, d_cnt, d_run, c1, dx, dy, (u32)c0, (u32)est, d_total + 1);
                    else if (segw) hipLaunchKernelGGL(k_sketch_compact<2>, cgrid, dim3(256), 0, ctx->stream, sx, sy, d_cnt, d_run, c1, dx, dy, (u32)c0, (u32)est, d_total + 1);
                    else hipLaunchKernelGGL(k_sketch_compact<true>, cgrid, dim3(256), 0, ctx->stream, sx, sy, d_cnt, d_run, c1, dx, dy, (u32)c0, (u32)est, d_total + 1);
                    KCHK(ctx);
                }
                if (gated) { int rr = seqset_ready(ctx, s); if (rr) return rr; gated = false; }
                hipLaunchKernelGGL(k_read_mz_offsets, dim3((u32)div_up((u64)s->n + 1, 256)), dim3(256), 0, ctx->stream, d_cs, d_cnt, s->n, n_chunks, d_run, d_mzoff);
                KCHK(ctx);
                u32 h_run = 0, h_ovf = 0;
                HIPCHK(ctx, ctx->d2h(&h_run, d_run, 4, ctx->stream));
                HIPCHK(ctx, ctx->d2h(&h_ovf, d_total + 1, 4, ctx->stream));
                if (h_mzoff) {
                    h_mzoff->resize((size_t)s->n + 1);
                    HIPCHK(ctx, ctx->d2h(h_mzoff->data(), d_mzoff, ((size_t)s->n + 1) * 4, ctx->stream));
                }
                HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
                sc.drop(rx); if (ry) sc.drop(ry); sc.drop(d_run);
                if (!h_ovf) {
                    sc.drop(d_cnt); sc.drop(d_total);
                    o->x = dx; o->y = dy; o->mz_off = d_mzoff; o->n = h_run; o->segw = segw;
                    return LRGE_OK;
                }
                sc.drop(dx); if (dy) sc.drop(dy);          // beat the estimate, or a chunk overflowed its slot: two passes
            } else {
                if (rx) sc.drop(rx); if (ry) sc.drop(ry); if (dx) sc.drop(dx); if (dy) sc.drop(dy); if (d_run) sc.drop(d_run);
                (void)hipGetLastError(); ctx->err.clear();
            }
        }
    }
    for (int pass = 0; pass < 2; ++pass) {       // second round only after a slot overflow
        HIPCHK(ctx, hipMemsetAsync(d_total, 0, 8, ctx->stream));
        if (n_chunks) {
            if (one_pass && gated && pass == 0) {
                // the set's upload is still in flight (host-side pack, chunk after chunk): the sketch chunks that lie wholly inside
                // the words of upload chunk j run behind gate j, while the later chunks are still being packed and sent
                u32 c_prev = 0;
                const size_t ng = gjob ? gjob->gate_w1.size() : 0;
                for (size_t j = 0; j < ng && c_prev < n_chunks; ++j) {
                    const u32 c_end = chunks_behind(gjob->gate_w1[j]);
                    if (c_end <= c_prev) continue;                               // (a gate in front of this view, or inside one long read)
                    if (!gjob->wait_gate((int)j)) break;                         // (the job failed: seqset_ready below reports it)
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, gjob->gate_ev[j], 0));
                    {
                        launch_slots(c_prev, c_end, tx, ty, sk_cap);
                        KCHK(ctx);
                        c_prev = c_end;
                    }
                }
                int rr = seqset_ready(ctx, s); if (rr) return rr;
                if (c_prev < n_chunks) {                                          // (whatever a failed / odd gate sequence left)
                    launch_slots(c_prev, n_chunks, tx, ty, sk_cap);
                    KCHK(ctx);
                }
            } else if (one_pass) {
                if (gated && pass == 0) { int rr = seqset_ready(ctx, s); if (rr) return rr; }
                launch_slots(0u, n_chunks, tx, ty, sk_cap);
            } else {
                if (gated && pass == 0) { int rr = seqset_ready(ctx, s); if (rr) return rr; }
                hipLaunchKernelGGL((k_sketch_count<K, W, HPC>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                                   n_chunks, d_cnt);
            }
            KCHK(ctx);
            int rc = scan_exclusive_u32(ctx, sc, d_cnt, d_cnt, n_chunks, d_total);
            if (rc) return rc;
        }
        // per-read offsets follow from the chunk scan alone: they travel to the host with the total, in the one sync
        hipLaunchKernelGGL(k_read_mz_offsets, dim3((u32)div_up((u64)s->n + 1, 256)), dim3(256), 0, ctx->stream, d_cs, d_cnt, s->n,
                           n_chunks, d_total, d_mzoff);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(tot_ovf, d_total, 8, ctx->stream));
        if (h_mzoff) {
            h_mzoff->resize((size_t)s->n + 1);
            HIPCHK(ctx, ctx->d2h(h_mzoff->data(), d_mzoff, ((size_t)s->n + 1) * 4, ctx->stream));
        }
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        if (!(one_pass && tot_ovf[1])) break;
        one_pass = false;                        // a chunk held more than SK_CAP minimizers: redo in two passes
        sc.drop(tx); if (ty) sc.drop(ty); tx = ty = nullptr;
    }
    const u32 total = tot_ovf[0];
    if (keep_slots && pk && one_pass && n_chunks && sk_cap == (u32)SK_CAP) {
        // no compaction: the caller's sort reads the slots (k_prims.h: radix_sort_keys_first_pass_from_slots)
        sc.drop(d_total);
        o->x = nullptr; o->y = nullptr; o->mz_off = d_mzoff; o->n = total;
        o->slots = tx; o->offs = d_cnt; o->n_chunks = n_chunks;
        return LRGE_OK;
    }
    ALLOC_OR_FAIL(dx, sc, u64, (size_t)total + 1);
    u64 *dy = nullptr;
    if (!pk) { dy = get_y((size_t)total + 1); if (!dy) return LRGE_ERR_DEVICE; }
    if (n_chunks && one_pass) {
        const dim3 cgrid((u32)div_up(div_up(n_chunks, 64), 4));
        if (pk) hipLaunchKernelGGL(k_sketch_compact<false>, cgrid, dim3(256), 0, ctx->stream, tx, ty, d_cnt, d_total, n_chunks, dx, dy);
        else if (segw) hipLaunchKernelGGL(k_sketch_compact<2>, cgrid, dim3(256), 0, ctx->stream, tx, ty, d_cnt, d_total, n_chunks, dx, dy);
        else hipLaunchKernelGGL(k_sketch_compact<true>, cgrid, dim3(256), 0, ctx->stream, tx, ty, d_cnt, d_total, n_chunks, dx, dy);
        KCHK(ctx);
        sc.drop(tx); if (ty) sc.drop(ty);
    } else if (n_chunks) {
        if (pk)
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, true, true>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, pk_pos1, pk_ybits);
        else if (segw)
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, true, 2>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, pk_pos1, pk_ybits);
        else if (index_keys)
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, true, false>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, 0u, 0u);
        else
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, false, false>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, 0u, 0u);
        KCHK(ctx);
    }
    // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
    sc.drop(d_cnt); sc.drop(d_total);
    o->x = dx; o->y = dy; o->mz_off = d_mzoff; o->n = total; o->segw = segw;
    return LRGE_OK;
}

static int sketch_device(lrge_hip_ctx *ctx, Scratch &sc, const lrge_hip_seqset *s, int preset, bool index_keys, SketchOut *o,
                         u32 pk_pos1 = 0, u32 pk_ybits = 0, std::vector<u32> *h_mzoff = nullptr, bool keep_slots = false, bool segw = false, bool wave_ok = false) {
    // A set whose host-side pack is still running on the uploader thread (chunk gates: host_pack.h) is sketched chunk by chunk
    // behind its transfer -- index sketches only (a streamed set's upload hides behind the index build anyway).  Since round 4 also
    // VIEWS of such a set (the parts of a partitioned index: part 0 is sketched, sorted and tabled while parts 1.. still travel)
    // and the HPC preset (whole reads only: an HPC step may read a homopolymer run past its chunk).  option NO_GATED_SKETCH: wait first.
    const lrge_hip_seqset *root_ = s->is_view ? s->view_root : s;
    const std::shared_ptr<UploadJob> &job_ = s->is_view ? s->view_job : s->job;
    const bool gated = index_keys && root_ && root_->pending && job_ && root_->job == job_ && !job_->gate_ev.empty() && s->n_words != 0 &&
                       (!s->is_view || s->view_gate >= 0) && !ctx->opt("NO_GATED_SKETCH") && s->n_chunks != 0 && s->n_chunks < (1ULL << 32);
    int rc = gated ? LRGE_OK : seqset_ready(ctx, s);
    if (rc) return rc;
    StageTimer t(ctx, LRGE_T_SKETCH);
    rc = (preset == LRGE_PRESET_AVA_PB) ? sketch_launch<19, 5, true>(ctx, sc, s, index_keys, o, pk_pos1, pk_ybits, h_mzoff, gated, keep_slots, segw, wave_ok)
                                            : sketch_launch<15, 5, false>(ctx, sc, s, index_keys, o, pk_pos1, pk_ybits, h_mzoff, gated, keep_slots, segw, wave_ok);
    t.stop();
    return rc;
}

// ---- presketch: the streamed set's minimizers, computed on the side stream with no host round trip ----
// Two steps, because the device arena recycles blocks in the order of the MAIN stream: everything the side stream will touch
// is allocated where it forks (presketch_prepare: nothing released by the index build after that point can be handed to it),
// the kernels may be queued later (presketch_launch_prepared).
static int presketch_alloc(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, PreSketch *p) {
    if (s->n_chunks >= (1ULL << 32) || s->total_bases + 1 >= (1ULL << 32)) return LRGE_ERR_TOO_MANY;
    Scratch &sc = *p->sc;
    const u64 nb = div_up(s->n_chunks, SCAN_TILE);
    if (nb > 8192) return LRGE_ERR_TOO_MANY;                   // (single-level scan with the caller's block sums)
    ALLOC_OR_FAIL(d_cnt, sc, u32, (size_t)s->n_chunks + 1);
    ALLOC_OR_FAIL(d_bs, sc, u32, (size_t)nb + 2);
    ALLOC_OR_FAIL(d_total, sc, u32, 1);
    ALLOC_OR_FAIL(d_mzoff, sc, u32, (size_t)s->n + 1);
    // the count is not known on the host when the write pass is queued: room for one minimizer per base
    ALLOC_OR_FAIL(dx, sc, u64, (size_t)s->total_bases + 1);
    ALLOC_OR_FAIL(dy, sc, u64, (size_t)s->total_bases + 1);
    p->cnt = d_cnt; p->bs = d_bs; p->x = dx; p->y = dy; p->mz_off = d_mzoff; p->d_total = d_total;
    return LRGE_OK;
}

template <int K, int W, bool HPC>
static int presketch_launch(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, PreSketch *p, hipStream_t st) {
    Scratch &sc = *p->sc;
    const u32 n_chunks = (u32)s->n_chunks;
    ChunkMap cm{s->d_cs, s->n};
    const dim3 sgrid((u32)div_up(n_chunks, SK_THREADS));
    // Two passes here, not the one-pass form of sketch_launch: beside the index's memory-bound passes a second VALU-bound pass
    // overlaps where the one-pass form's streaming compaction competes.  Measured twice: beside the sort passes (round 2: the sort
    // loses what the sketch gains) and beside the table build, where this runs now (round 3: C4 step 32.7-32.8 ms with two passes,
    // 33.1-33.4 with slots + compaction on the same box).
    if (n_chunks) {
        hipLaunchKernelGGL((k_sketch_count<K, W, HPC>), sgrid, dim3(SK_THREADS), 0, st, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, p->cnt);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, p->cnt, p->cnt, n_chunks, p->d_total, st, true, p->bs);
        if (rc) return rc;
    } else {
        HIPCHK(ctx, hipMemsetAsync(p->d_total, 0, 4, st));
    }
    hipLaunchKernelGGL(k_read_mz_offsets, dim3((u32)div_up((u64)s->n + 1, 256)), dim3(256), 0, st, s->d_cs, p->cnt, s->n, n_chunks, p->d_total,
                       p->mz_off);
    KCHK(ctx);
    if (n_chunks) {
        hipLaunchKernelGGL((k_sketch_write<K, W, HPC, false, false>), sgrid, dim3(SK_THREADS), 0, st, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                           n_chunks, p->cnt, p->x, p->y, 0u, 0u);
        KCHK(ctx);
    }
    return LRGE_OK;
}

static void presketch_drop_prepared(lrge_hip_ctx *ctx) {
    PreSketch *p = ctx->presk_prepared;
    if (!p) return;
    ctx->presk_prepared = nullptr; ctx->presk_prepared_set = nullptr;
    delete p->sc;                                        // (nothing has been queued on these blocks)
    ctx->event_pool.push_back(p->ev_start); ctx->event_pool.push_back(p->ev_done);
    delete p;
}

// Called by the index build right after its own sketch has been queued on ctx->stream: marks the point of the main stream
// the side stream starts from and takes the memory of the streamed set's sketch.
// indexed_bases: size of the set whose index build would hide the sketch.  A streamed set several times larger than the
// indexed one (the inverse strategy on a big job: 3 Gbases streamed against a 150 Mbase index) finds nothing to hide behind --
// the two VALU-bound sketches and the small sort just share the chip -- so the hint is ignored there and the overlap call
// sketches in line (C5/10 inverse: 95 -> 89 ms per step).
static int presketch_prepare(lrge_hip_ctx *ctx, u64 indexed_bases) {
    presketch_drop_prepared(ctx);
    lrge_hip_seqset *s = ctx->presk_pending;
    if (!s) return LRGE_OK;
    ctx->presk_pending = nullptr;
    if (s->total_bases > 2 * indexed_bases && !ctx->opt("PRESKETCH_ALWAYS")) return LRGE_OK;
    if (s->total_bases > ctx->opt_u64("STREAM_BASES", 4000000000ull)) return LRGE_OK;   // streamed in views: sketched per view
    if (s->presk) presketch_discard(s);
    PreSketch *p = new PreSketch();
    p->preset = ctx->presk_preset;
    p->sc = new Scratch(ctx);
    p->ev_start = ctx->get_event(); p->ev_done = ctx->get_event();
    ctx->presk_prepared = p; ctx->presk_prepared_set = s;
    // behind the index sketch (both are VALU-bound; the point is to run beside the passes that follow it)
    if (presketch_alloc(ctx, s, p) != LRGE_OK || hipEventRecord(ctx->ev_presk, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        presketch_drop_prepared(ctx);                    // not fatal: the overlap call sketches the set itself
    }
    return LRGE_OK;
}

// Queues the prepared sketch on the side stream.  May block on the HOST until the set's upload job (host-side pack) is over,
// which is why the index build calls it only once it has nothing more of its own to queue that could run meanwhile.
static int presketch_launch_prepared(lrge_hip_ctx *ctx) {
    PreSketch *p = ctx->presk_prepared; lrge_hip_seqset *s = ctx->presk_prepared_set;
    if (!p) return LRGE_OK;
    hipError_t e = hipStreamWaitEvent(ctx->stream2, ctx->ev_presk, 0);
    // an upload of the set still in flight: only the side stream waits for it -- the main stream goes on with the index
    // (its own seqset_ready comes with the overlap call, which also returns the staging blocks to the pool)
    if (s->job && seqset_job_wait(ctx, s) != LRGE_OK) { presketch_drop_prepared(ctx); return LRGE_OK; }
    if (e == hipSuccess && s->pending) e = hipStreamWaitEvent(ctx->stream2, s->ev_ready, 0);
    if (e == hipSuccess) e = hipEventRecord(p->ev_start, ctx->stream2);
    int rc = LRGE_OK;
    if (e == hipSuccess) {
        rc = p->preset == LRGE_PRESET_AVA_PB ? presketch_launch<19, 5, true>(ctx, s, p, ctx->stream2)
                                             : presketch_launch<15, 5, false>(ctx, s, p, ctx->stream2);
        if (rc == LRGE_OK) e = hipEventRecord(p->ev_done, ctx->stream2);
    }
    if (e != hipSuccess || rc != LRGE_OK) {      // not fatal: the overlap call sketches the set itself
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipGetLastError();
        presketch_drop_prepared(ctx);
        return LRGE_OK;
    }
    ctx->presk_prepared = nullptr; ctx->presk_prepared_set = nullptr;
    s->presk = p;
    return LRGE_OK;
}

// may_block = false (one part of a partitioned index): a streamed set whose upload job is still running -- it is queued behind
// the targets' own on the uploader thread, i.e. behind parts that have not even arrived -- keeps its request for the next part's
// build instead of stalling the host (with this part's table passes unqueued) until the whole transfer is over
static int presketch_start_pending(lrge_hip_ctx *ctx, u64 indexed_bases, bool may_block = true) {
    if (!may_block && ctx->presk_pending && ctx->presk_pending->job && !ctx->presk_pending->job->is_done()) return LRGE_OK;
    int rc = presketch_prepare(ctx, indexed_bases);
    return rc ? rc : presketch_launch_prepared(ctx);
}

extern "C" int lrge_hip_seqset_presketch(lrge_hip_ctx *ctx, lrge_hip_seqset *s, int preset) {
    if (!ctx || !s || s->ctx != ctx) return LRGE_ERR_INVALID;
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    ctx->presk_pending = s; ctx->presk_preset = preset;
    return LRGE_OK;
}

extern "C" int lrge_hip_sketch_dump(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, int preset, uint64_t *x, uint64_t *y,
                                    uint64_t cap, uint64_t *n_out) {
    if (!ctx || !s || !n_out) return LRGE_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    Scratch sc(ctx);
    SketchOut o;
    int rc = sketch_device(ctx, sc, s, preset, false, &o);
    if (rc) return rc;
    *n_out = o.n;
    u64 m = o.n < cap ? o.n : cap;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
    if (m && x) HIPCHK(ctx, hipMemcpy(x, o.x, m * 8, hipMemcpyDeviceToHost));
    if (m && y) HIPCHK(ctx, hipMemcpy(y, o.y, m * 8, hipMemcpyDeviceToHost));
    return LRGE_OK;
}
