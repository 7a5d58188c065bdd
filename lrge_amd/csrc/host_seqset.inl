// host_seqset.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): read sets: upload (device pack, host pack + background uploader), views, presketch hand-over, release.
// ------------------------------------------------------------------------------------------
// read sets
// ------------------------------------------------------------------------------------------
// Name ranks are positions in the sorted union of the names that meet in a call, i.e. small dense integers: duplicates and
// intersections are found with one bitmap pass instead of a sort per upload (a sort of 100 000 ranks was ~1 ms of host time
// in front of every index build).  Sparse rank values (a caller's own numbering) fall back to sorting.
static bool ranks_have_duplicate(const std::vector<u32> &r) {
    if (r.size() < 2) return false;
    u32 mx = 0;
    for (u32 v : r) mx = v > mx ? v : mx;
    if ((u64)mx <= 64ull * r.size() + 1024) {
        std::vector<u64> bits(((size_t)mx >> 6) + 1, 0);
        for (u32 v : r) { u64 &w = bits[v >> 6]; const u64 m = 1ULL << (v & 63); if (w & m) return true; w |= m; }
        return false;
    }
    std::vector<u32> t(r);
    std::sort(t.begin(), t.end());
    for (size_t i = 1; i < t.size(); ++i) if (t[i] == t[i - 1]) return true;
    return false;
}
static bool ranks_intersect(const std::vector<u32> &a, const std::vector<u32> &b) {
    if (a.empty() || b.empty()) return false;
    const std::vector<u32> &small = a.size() <= b.size() ? a : b, &large = a.size() <= b.size() ? b : a;
    u32 mx = 0;
    for (u32 v : large) mx = v > mx ? v : mx;
    if ((u64)mx <= 64ull * large.size() + 1024) {
        std::vector<u64> bits(((size_t)mx >> 6) + 1, 0);
        for (u32 v : large) bits[v >> 6] |= 1ULL << (v & 63);
        for (u32 v : small) if (v <= mx && (bits[v >> 6] >> (v & 63)) & 1) return true;
        return false;
    }
    std::vector<u32> t(large);
    std::sort(t.begin(), t.end());
    for (u32 v : small) if (std::binary_search(t.begin(), t.end(), v)) return true;
    return false;
}

extern "C" int lrge_hip_host_alloc(size_t bytes, void **out) {
    if (!out) return LRGE_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); g_last_error = hipGetErrorString(e); return LRGE_ERR_DEVICE; }
    return LRGE_OK;
}
extern "C" void lrge_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

// a host-side pack still running on the uploader thread: its last act is to record the set's ev_ready, so nobody may wait
// for that event on the device before the job has finished on the host
static int seqset_job_wait(lrge_hip_ctx *ctx, lrge_hip_seqset *s) {
    if (!s->job) return LRGE_OK;
    std::string e;
    const int rc = s->job->wait(&e);
    for (hipEvent_t g : s->job->gate_ev) ctx->event_pool.push_back(g);
    s->job->gate_ev.clear();
    s->job.reset();
    if (rc) { ctx->err = e; return rc; }
    return LRGE_OK;
}

// Every consumer of a set's device arrays calls this first: work queued on the main stream after it runs behind the
// set's upload; the staging blocks of the upload return to the pool (recycled in main-stream order from here on).
static int seqset_ready(lrge_hip_ctx *ctx, const lrge_hip_seqset *cs) {
    lrge_hip_seqset *s = const_cast<lrge_hip_seqset *>(cs);
    if (s->is_view) {
        if (s->view_gate < 0) return LRGE_OK;
        lrge_hip_seqset *root = s->view_root;
        if (root->job && root->job == s->view_job) {
            // the parent's upload is (or was) running: wait for the chunk that carries this view's last word -- on the host until its
            // transfer has been queued, on the device for the transfer itself (the per-read arrays went ahead of the first chunk)
            if (!s->view_job->wait_gate(s->view_gate)) { const int jrc = seqset_job_wait(ctx, root); return jrc ? jrc : LRGE_ERR_DEVICE; }
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->view_job->gate_ev[(size_t)s->view_gate], 0));
        } else if (root->pending) {
            // somebody has consumed the job object since (its gate events are recycled): the set's own event covers everything
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, root->ev_ready, 0));
        }
        s->view_gate = -1; s->view_job.reset();
        return LRGE_OK;
    }
    if (!s->pending) return LRGE_OK;
    { int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_ready, 0));
    s->pending = false;
    if (s->meta_arena) { s->meta_arena = false; if (--ctx->meta_inflight == 0) ctx->meta_used = 0; }
    ctx->pool.release(s->stg_ascii);      // (an arena block: whatever its size, it serves any later request)
    s->stg_ascii = nullptr;       // (stg_boff / stg_blk live inside the set's meta block)
    return LRGE_OK;
}

// pageable source -> pinned staging buffer with a few host threads (one thread moves ~10 GB/s, PCIe Gen5 x16 ~55)
static void parallel_memcpy(char *dst, const char *src, size_t n) {
    const size_t kMin = (size_t)4 << 20;
    const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, n / kMin));
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / nt + 63) & ~(size_t)63;
    for (unsigned t = 1; t < nt; ++t) {
        const size_t o = std::min(n, per * t), e = std::min(n, per * (t + 1));
        if (e > o) th.emplace_back([=] { memcpy(dst + o, src + o, e - o); });
    }
    memcpy(dst, src, std::min(n, per));
    for (auto &x : th) x.join();
}

#define PACK_MIN_CPUS_PER_RANK 8.0
// true: pack on the host (AVX2, host_pack.h) and send packed words; false: send the ASCII and run k_pack.  One rank on the host: always the
// host (measured best: 0.375 B per base over PCIe instead of 1).  Several: only when every rank has PACK_MIN_CPUS_PER_RANK CPUs of its own.
static bool pack_on_host(const char *opt, int ranks_on_host, double *granted) {
    const double q = hp_cpu_quota();
    if (granted) *granted = q;
    if (opt && !strcmp(opt, "device")) return false;
    if (opt && !strcmp(opt, "host")) return true;
    if (ranks_on_host <= 1) return true;
    return q / (double)ranks_on_host >= PACK_MIN_CPUS_PER_RANK;
}
extern "C" int lrge_hip_pack_choice(int ranks_on_host, double *granted_cpus) {
    return pack_on_host(nullptr, ranks_on_host, granted_cpus) ? 1 : 0;
}

static int seqset_upload_impl(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n, const uint32_t *name_rank,
                              bool async, lrge_hip_seqset **out) {
    if (!ctx || !out || (n && (!bases || !offsets))) return LRGE_ERR_INVALID;
    *out = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    std::unique_ptr<lrge_hip_seqset, void (*)(lrge_hip_seqset *)> guard(new lrge_hip_seqset(), lrge_hip_seqset_free);
    lrge_hip_seqset *s = guard.get();
    s->ctx = ctx; s->n = n; s->pooled = true; s->uid = g_seqset_uid.fetch_add(1);
    // one pass over the offsets: word offsets, lengths, sketch chunk map (read -> first chunk, fixed for the life of the set)
    s->h_woff.resize((size_t)n + 1); s->h_len.resize(n ? n : 1); s->h_cs.resize((size_t)n + 1);
    u64 w = 0, nc = 0;
    {
        u64 *hw = s->h_woff.data(); u32 *hl = s->h_len.data(), *hc = s->h_cs.data();
        u32 max_len = 0; bool has_empty = false;
        for (u32 i = 0; i < n; ++i) {
            const u64 d = offsets[i + 1] - offsets[i];
            if (offsets[i + 1] < offsets[i] || d >= (1ULL << 31)) {
                LRGE_SET_ERR(ctx, "read %u: bad offsets or length >= 2^31", i); return LRGE_ERR_INVALID;
            }
            const u32 len = (u32)d;
            hw[i] = w; hl[i] = len; hc[i] = (u32)nc;
            w += (len + 31) / 32; nc += (len + SK_CHUNK - 1) / SK_CHUNK;
            has_empty |= len == 0;
            max_len = len > max_len ? len : max_len;
        }
        hw[n] = w; hc[n] = (u32)nc;
        s->max_len = max_len; s->has_empty = has_empty;
    }
    s->n_words = w; s->n_chunks = nc;
    s->total_bases = n ? offsets[n] - offsets[0] : 0;
    if (name_rank) {
        s->has_rank = true;
        s->h_rank.assign(name_rank, name_rank + n);
        s->dup_rank = ranks_have_duplicate(s->h_rank);
    }
    const u64 n_blk = div_up(w, PACK_WORDS);
    hipError_t e = hipSuccess;
    auto alloc = [&](size_t bytes) -> void * { return ctx->pool.alloc(bytes, &e); };
    const size_t nw = (size_t)(w ? w : 1);
    s->d_pack = (u64 *)alloc(nw * 8); s->d_nmask = (u32 *)alloc(nw * 4);
    // the per-read arrays: one device block, one host image, one transfer
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_woff = 0, o_boff = o_woff + al(((size_t)n + 1) * 8), o_blk = o_boff + al(((size_t)n + 1) * 8);
    const size_t o_cs = o_blk + al((size_t)(n_blk + 1) * 4), o_len = o_cs + al(((size_t)n + 1) * 4);
    const size_t o_rank = o_len + al((size_t)(n ? n : 1) * 4), meta_bytes = o_rank + al((size_t)(n ? n : 1) * 4);
    s->d_meta = alloc(meta_bytes);
    if (!s->d_pack || !s->d_nmask || !s->d_meta) {
        LRGE_SET_ERR(ctx, "seqset_upload: device allocation failed: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE;
    }
    char *dm = (char *)s->d_meta;
    s->d_woff = (u64 *)(dm + o_woff); s->stg_boff = dm + o_boff; s->stg_blk = dm + o_blk;
    s->d_cs = (u32 *)(dm + o_cs); s->d_len = (u32 *)(dm + o_len); s->d_rank = (u32 *)(dm + o_rank);
    // where do the bases live?  device memory (no copy at all), pinned host memory (one DMA), pageable host memory (staged)
    const char *src = n ? bases + offsets[0] : nullptr;
    int kind = 2;                                         // 0 device, 1 pinned host, 2 pageable host
    if (src && s->total_bases) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice) kind = 0; else if (at.type == hipMemoryTypeHost) kind = 1;
        } else (void)hipGetLastError();
    }
    const u8 *d_ascii = (const u8 *)src;
    // a set that starts in host memory is packed on the host and travels packed (host_pack.h); option NO_HOST_PACK sends the
    // ASCII and packs on the device as rounds 1-2 did
    // Which side packs (round 6): the host-side pack needs CPUs -- 16 granted CPUs pack H. sapiens scale in 258 ms whatever the number of
    // ranks (profiles/r05_pack_contention.json) -- so a rank that is granted fewer than PACK_MIN_CPUS_PER_RANK (8) of them ships the ASCII over
    // its own PCIe link and runs k_pack (option PACK = host | device | auto; RANKS_ON_HOST = the ranks sharing this host's CPUs, set by the
    // launcher: lrge_hip_pack_choice states the rule)
    const bool host_pack = kind != 0 && s->total_bases > 0 && !ctx->opt("NO_HOST_PACK") && pack_on_host(ctx->opt("PACK"), (int)ctx->opt_u64("RANKS_ON_HOST", 1), nullptr);
    if (kind != 0 && s->total_bases && !host_pack) {
        s->stg_ascii = alloc(s->total_bases);
        if (!s->stg_ascii) { LRGE_SET_ERR(ctx, "seqset_upload: device allocation failed: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE; }
        d_ascii = (const u8 *)s->stg_ascii;
    }
    if (!s->ev_ready) s->ev_ready = ctx->get_event();
    hipStream_t cs = ctx->copy_stream;
    // the blocks just taken from the pool may still be in use by work queued on the main stream
    HIPCHK(ctx, hipEventRecord(ctx->ev_gate, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(cs, ctx->ev_gate, 0));
    s->pending = true;                                     // (from here on seqset_free drains the copy stream first)
    {
        if (!ctx->meta_pin && hipHostMalloc((void **)&ctx->meta_pin, (size_t)32 << 20, hipHostMallocDefault) == hipSuccess) ctx->meta_cap = (size_t)32 << 20;
        else if (!ctx->meta_pin) (void)hipGetLastError();
        char *hm = nullptr;
        if (ctx->meta_pin && ctx->meta_used + meta_bytes <= ctx->meta_cap) {
            // a rewound arena: every consumer of the earlier uploads has ordered itself behind them on the DEVICE
            // (seqset_ready); the host must not overwrite the bytes before the last transfer has actually read them
            // (it almost always has: ~2 us)
            if (ctx->meta_used == 0 && ctx->ev_meta) HIPCHK(ctx, hipEventSynchronize(ctx->ev_meta));
            hm = ctx->meta_pin + ctx->meta_used; ctx->meta_used += meta_bytes; ++ctx->meta_inflight; s->meta_arena = true;
        }
        auto put = [&](size_t off, const void *src_, size_t bytes) -> hipError_t {
            if (hm) { memcpy(hm + off, src_, bytes); return hipSuccess; }
            return hipMemcpyAsync(dm + off, src_, bytes, hipMemcpyHostToDevice, cs);       // (arena full: piecewise, from the set's own vectors)
        };
        HIPCHK(ctx, put(o_woff, s->h_woff.data(), ((size_t)n + 1) * 8));
        {   // k_pack's two maps exist for the upload only: base offset of every read relative to the first, and the read that
            // holds the first word of every block -- produced where they travel from (the arena; the set's own vectors when
            // it is full, because an asynchronous copy reads them after this call has returned)
            u64 *boff; u32 *blk;
            if (hm) { boff = (u64 *)(hm + o_boff); blk = (u32 *)(hm + o_blk); }
            else { s->h_boff.resize((size_t)n + 1); s->h_blk.resize((size_t)n_blk + 1); boff = s->h_boff.data(); blk = s->h_blk.data(); }
            const u64 o0 = n ? offsets[0] : 0;
            for (u32 i = 0; i <= n; ++i) boff[i] = n ? offsets[i] - o0 : 0;
            const u64 *hw = s->h_woff.data();
            u32 r = 0;
            for (u64 bq = 0; bq < n_blk; ++bq) {
                const u64 w0 = bq * PACK_WORDS;
                while (r + 1 < n && hw[r + 1] <= w0) ++r;
                blk[bq] = r;
            }
            blk[n_blk] = 0;
            if (!hm) {
                HIPCHK(ctx, put(o_boff, boff, ((size_t)n + 1) * 8));
                HIPCHK(ctx, put(o_blk, blk, (size_t)(n_blk + 1) * 4));
            }
        }
        if (s->n_chunks < (1ULL << 32)) HIPCHK(ctx, put(o_cs, s->h_cs.data(), ((size_t)n + 1) * 4));
        if (n) {
            HIPCHK(ctx, put(o_len, s->h_len.data(), (size_t)n * 4));
            if (name_rank) HIPCHK(ctx, put(o_rank, s->h_rank.data(), (size_t)n * 4));
        }
        if (hm) {
            HIPCHK(ctx, hipMemcpyAsync(dm, hm, meta_bytes, hipMemcpyHostToDevice, cs));
            if (!ctx->ev_meta) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_meta, hipEventDisableTiming));
            HIPCHK(ctx, hipEventRecord(ctx->ev_meta, cs));
        }
    }
    if (host_pack) {
        // pinned chunk buffers + the uploader thread, once per context
        // words per chunk: 64 Mbases for sets up to 4 Gbases (the first chunk is what the index sketch waits for: 0.9 ms at C4), 1/64
        // of the set above that, at most 512 Mbases -- every chunk costs the uploader a round of worker wake-ups and two copies, and
        // the 447 chunks of full-size C5's targets took 370 ms where 64 take 270 (measured: tools/sweeps/r4_pack_sweep.sh)
        const size_t CH = (size_t)ctx->opt_u64("HOST_PACK_CHUNK_WORDS", std::min<u64>((u64)16 << 20, std::max<u64>((u64)2 << 20, (w / 64 + 65535) & ~65535ull)));
        if (!ctx->hp_stage[0] || ctx->hp_words < CH) {          // (hp_words: the buffers' capacity; a job lays its own chunk size out in them)
            if (ctx->uploader) ctx->uploader->drain();        // an earlier upload's job may still pack into the buffers about to go (it captured them and CH by value)
            for (int b = 0; b < 2; ++b) {
                if (ctx->hp_stage[b]) { HIPCHK(ctx, hipStreamSynchronize(cs)); (void)hipHostFree(ctx->hp_stage[b]); ctx->hp_stage[b] = nullptr; }
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->hp_stage[b], CH * 12, hipHostMallocDefault));
                if (!ctx->hp_ev[b]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->hp_ev[b], hipEventDisableTiming));
                HIPCHK(ctx, hipEventRecord(ctx->hp_ev[b], cs));
            }
            ctx->hp_words = CH;
        }
        if (!ctx->uploader) {
            ctx->uploader = new Uploader();
            if (!ctx->opt("HOST_PACK_NO_PIN")) ctx->uploader->cpus = hp_gpu_node_cpus(ctx->device, &ctx->uploader->node);     // the GPU's own NUMA node
            // threads packing at once = what the host grants this process (hp_cpu_quota: affinity and cgroup bandwidth), less two for
            // the thread that launches kernels and the runtime's own; at most 32, at most half the node's hardware threads.  The FIRST
            // context of the process on a NUMA node sizes the pool (HOST_PACK_THREADS overrides); several PROCESSES on one host (one per
            // GPU over RCCL) share its CPUs without knowing of each other: their launcher divides (bench.py sets HOST_PACK_THREADS)
            const u32 hw = ctx->uploader->cpus.empty() ? std::max(2u, std::thread::hardware_concurrency()) : (u32)ctx->uploader->cpus.size() * 2;
            const double quota = hp_cpu_quota();
            const u32 dflt = std::min<u32>(std::min<u32>(32, std::max<u32>(2, hw / 2)), (u32)std::max(2.0, quota - 2.0));
            ctx->uploader->pool = hp_shared_pool(ctx->uploader->node, (u32)std::max<u64>(1, ctx->opt_u64("HOST_PACK_THREADS", dflt)) - 1, ctx->uploader->cpus);    // (>= 1: the uploader thread itself packs; a CPU of its own per worker instead of the node: measured, no difference)
            if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] host-side pack: %zu worker threads (shared by the process) on %zu CPUs of the GPU's NUMA node; the host grants %.1f CPUs\n", ctx->uploader->pool->th.size(), ctx->uploader->cpus.size(), quota);
        }
        s->h_boff.resize((size_t)n + 1);
        { const u64 o0 = offsets[0]; for (u32 i = 0; i <= n; ++i) s->h_boff[i] = offsets[i] - o0; }
        auto job = std::make_shared<UploadJob>();
        s->job = job;
        for (u64 w0 = 0; w0 < w; w0 += CH) { job->gate_ev.push_back(ctx->get_event()); job->gate_w1.push_back(std::min<u64>(w, w0 + CH)); }
        hipEvent_t ev_ready = s->ev_ready;
        const int device = ctx->device;
        const u64 n_words = w;
        u64 *d_pack = s->d_pack; u32 *d_nmask = s->d_nmask;
        const u64 *boff = s->h_boff.data(), *woff = s->h_woff.data();
        Uploader *up = ctx->uploader;
        char **stage = ctx->hp_stage; hipEvent_t *sev = ctx->hp_ev;
        const u8 *hsrc = (const u8 *)src;
        const bool verbose = ctx->opt("VERBOSE") != nullptr;
        auto work = [=]() {
            hipError_t e = hipSetDevice(device);
            int b = 0;
            const double t_job = DevPool::now_ms(); double t_pack = 0, t_wait = 0;
            for (u64 w0 = 0; w0 < n_words && e == hipSuccess; w0 += CH, b ^= 1) {
                const u64 w1 = std::min<u64>(n_words, w0 + CH), nw = w1 - w0;
                const double t0 = DevPool::now_ms();
                e = hipEventSynchronize(sev[b]);                      // the DMA that last read this buffer
                if (e != hipSuccess) break;
                const double t1 = DevPool::now_ms(); t_wait += t1 - t0;
                u64 *hp = (u64 *)stage[b]; u32 *hm = (u32 *)(stage[b] + CH * 8);
                const u32 n_tasks = (u32)std::min<u64>(256, std::max<u64>(1, nw / 16384));
                up->pool->parallel_for(n_tasks, [=](u32 t) {
                    const u64 a = w0 + nw * t / n_tasks, z = w0 + nw * (t + 1) / n_tasks;
                    hp_pack_range(hsrc, boff, woff, n, a, z, hp + (a - w0), hm + (a - w0));
                });
                t_pack += DevPool::now_ms() - t1;
                e = hipMemcpyAsync(d_pack + w0, hp, nw * 8, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(d_nmask + w0, hm, nw * 4, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipEventRecord(sev[b], cs);
                if (e == hipSuccess) { e = hipEventRecord(job->gate_ev[(size_t)(w0 / CH)], cs); if (e == hipSuccess) job->gate_recorded(); }
            }
            if (e == hipSuccess) e = hipEventRecord(ev_ready, cs);
            if (verbose) fprintf(stderr, "[lrge_hip] host-side pack of %llu words: job %.2f ms on the uploader thread (packing %.2f ms, waiting for a chunk buffer %.2f ms)\n",
                                 (unsigned long long)n_words, DevPool::now_ms() - t_job, t_pack, t_wait);
            job->finish(e == hipSuccess ? LRGE_OK : LRGE_ERR_DEVICE, e == hipSuccess ? std::string() : std::string("host-side pack / upload: ") + hipGetErrorString(e));
        };
        // a pinned source stays valid until the set is consumed (the contract of the async form): the job runs in the
        // background.  A pageable source may change as soon as this call returns, and the blocking form waits anyway.
        if (async && kind == 1) up->submit(work);
        else { up->submit(work); const int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    } else if (kind == 1) {
        HIPCHK(ctx, hipMemcpyAsync(s->stg_ascii, src, s->total_bases, hipMemcpyHostToDevice, cs));
    } else if (kind == 2 && s->total_bases) {
        if (!ctx->stage_cap) {       // both buffers and both events, or nothing (a half-made pair would fail every later upload)
            const size_t cap = (size_t)64 << 20;
            char *bufs[2] = {nullptr, nullptr}; hipEvent_t evs[2] = {nullptr, nullptr};
            hipError_t se = hipSuccess;
            for (int b = 0; b < 2 && se == hipSuccess; ++b) {
                se = hipHostMalloc((void **)&bufs[b], cap, hipHostMallocDefault);
                if (se == hipSuccess) se = hipEventCreateWithFlags(&evs[b], hipEventDisableTiming);
                if (se == hipSuccess) se = hipEventRecord(evs[b], cs);
            }
            if (se != hipSuccess) {
                for (int b = 0; b < 2; ++b) { if (bufs[b]) (void)hipHostFree(bufs[b]); if (evs[b]) (void)hipEventDestroy(evs[b]); }
                (void)hipGetLastError();
                LRGE_SET_ERR(ctx, "seqset_upload: pinned staging buffers: %s", hipGetErrorString(se));
                return LRGE_ERR_DEVICE;
            }
            for (int b = 0; b < 2; ++b) { ctx->stage[b] = bufs[b]; ctx->stage_ev[b] = evs[b]; }
            ctx->stage_cap = cap;
        }
        int b = 0;
        for (u64 o = 0; o < s->total_bases; o += ctx->stage_cap, b ^= 1) {
            const size_t len = (size_t)std::min<u64>(ctx->stage_cap, s->total_bases - o);
            HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[b]));          // the DMA that last read this buffer
            parallel_memcpy(ctx->stage[b], src + o, len);
            HIPCHK(ctx, hipMemcpyAsync((char *)s->stg_ascii + o, ctx->stage[b], len, hipMemcpyHostToDevice, cs));
            HIPCHK(ctx, hipEventRecord(ctx->stage_ev[b], cs));
        }
    }
    if (w && !host_pack) {
        // (timed only in the blocking form: a pending event pair would make the next call's timer resolution wait for
        // this upload on the host)
        std::unique_ptr<StageTimer> t(async ? nullptr : new StageTimer(ctx, LRGE_T_PACK, cs));
        hipLaunchKernelGGL(k_pack, dim3((u32)n_blk), dim3(PACK_THREADS), 0, cs, d_ascii, (const u64 *)s->stg_boff, s->d_woff,
                           (const u32 *)s->stg_blk, n, w, s->d_pack, s->d_nmask);
        KCHK(ctx);
    }
    if (!host_pack) HIPCHK(ctx, hipEventRecord(s->ev_ready, cs));      // (a host-side pack records it at the end of its job)
    // async: the per-read arrays travel from the set's own host copies (they live as long as the set); only `bases`
    // must stay valid, and only when it is pinned host memory (a pageable source has been copied out by now)
    if (!async) {
        HIPCHK(ctx, hipStreamSynchronize(cs));
        ctx->resolve_timers();
        // the set is complete: its 1 B/base ASCII staging block goes back now, not when somebody consumes the set
        s->pending = false;
        if (s->meta_arena) { s->meta_arena = false; if (--ctx->meta_inflight == 0) ctx->meta_used = 0; }
        ctx->pool.release(s->stg_ascii); s->stg_ascii = nullptr;
    }
    if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] upload of %u reads: %.3f ms of host time\n", n,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    *out = guard.release();
    return LRGE_OK;
}

extern "C" int lrge_hip_seqset_upload(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n,
                                      const uint32_t *name_rank, lrge_hip_seqset **out) {
    return seqset_upload_impl(ctx, bases, offsets, n, name_rank, false, out);
}
extern "C" int lrge_hip_seqset_upload_async(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n,
                                            const uint32_t *name_rank, lrge_hip_seqset **out) {
    return seqset_upload_impl(ctx, bases, offsets, n, name_rank, true, out);
}
extern "C" int lrge_hip_seqset_wait(lrge_hip_seqset *s) {
    if (!s) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    if (s->pending) HIPCHK(ctx, hipEventSynchronize(s->ev_ready));
    return LRGE_OK;
}

static void presketch_drop_prepared(lrge_hip_ctx *ctx);
static void presketch_discard(lrge_hip_seqset *s) {
    lrge_hip_ctx *ctx = s->ctx;
    if (ctx->presk_pending == s) ctx->presk_pending = nullptr;
    if (ctx->presk_prepared_set == s) presketch_drop_prepared(ctx);
    if (!s->presk) return;
    (void)hipStreamSynchronize(ctx->stream2);          // its kernels may still be running
    delete s->presk->sc;
    ctx->event_pool.push_back(s->presk->ev_start); ctx->event_pool.push_back(s->presk->ev_done);
    delete s->presk;
    s->presk = nullptr;
}

extern "C" void lrge_hip_seqset_free(lrge_hip_seqset *s) {
    if (!s) return;
    bool ctx_alive;
    { std::lock_guard<std::mutex> g(g_live_mu); ctx_alive = g_live_ctx.count(s->ctx) != 0; }
    if (ctx_alive) { (void)hipSetDevice(s->ctx->device); presketch_discard(s); }   // (a set that outlives its context only owns its own arrays)
    if (s->is_view) {      // a view owns its chunk map only (a pool block: hipFree would synchronise the device, i.e. wait for an upload in flight)
        if (ctx_alive) s->ctx->pool.release(s->d_cs);
        delete s; return;
    }
    if (s->pooled) {
        if (ctx_alive) {       // (a destroyed context has already freed its pool)
            lrge_hip_ctx *ctx = s->ctx;
            if (s->job) (void)seqset_job_wait(ctx, s);
            if (s->pending) (void)hipStreamSynchronize(ctx->copy_stream);          // an upload nobody consumed
            DevPool &P = ctx->pool;
            if (s->meta_arena && --ctx->meta_inflight == 0) ctx->meta_used = 0;
            P.release(s->d_pack); P.release(s->d_nmask); P.release(s->d_meta); P.release(s->stg_ascii);
            if (s->ev_ready) ctx->event_pool.push_back(s->ev_ready);
        }
    } else {
        (void)hipFree(s->d_pack); (void)hipFree(s->d_nmask); (void)hipFree(s->d_woff); (void)hipFree(s->d_len); (void)hipFree(s->d_rank); (void)hipFree(s->d_cs);
    }
    delete s;
}
extern "C" uint32_t lrge_hip_seqset_size(const lrge_hip_seqset *s) { return s ? s->n : 0; }
