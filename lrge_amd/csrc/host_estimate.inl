// host_estimate.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): per-read estimates, quantiles / median, and the seeded sub-sampling helpers.
// ------------------------------------------------------------------------------------------
// estimates
// ------------------------------------------------------------------------------------------
extern "C" int lrge_hip_estimates(lrge_hip_ctx *ctx, const uint32_t *counts, const uint32_t *read_lens, uint32_t n,
                                  float avg_target_len, uint64_t n_target_reads, uint32_t overlap_thresh, float *out) {
    if (!ctx || (n && (!counts || !read_lens || !out))) return LRGE_ERR_INVALID;
    if (n == 0) return LRGE_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    Scratch sc(ctx);
    ALLOC_OR_FAIL(dc, sc, u32, n); ALLOC_OR_FAIL(dl, sc, u32, n); ALLOC_OR_FAIL(d_out, sc, float, n);
    HIPCHK(ctx, hipMemcpyAsync(dc, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dl, read_lens, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    // `n_target_reads as f32`, `2.0 * ovlap_thresh as f32` (estimate.rs:153-156)
    float nt = (float)n_target_reads, two_thr = 2.0f * (float)overlap_thresh;
    hipLaunchKernelGGL(k_estimate, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx->stream, dc, dl, n, avg_target_len, nt, two_thr, d_out);
    KCHK(ctx);
    HIPCHK(ctx, ctx->d2h(out, d_out, (size_t)n * 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    return LRGE_OK;
}

// estimate.rs:80-132.  f32 arithmetic, no contraction (this TU is built with -ffp-contract=off).  `d` must hold the
// order statistics idx and idx + 1 at their sorted positions (the caller selects them; a full sort is not needed).
static size_t quantile_index(size_t n, float q) {
    volatile float pos = q * (float)(n - 1);
    return (size_t)floorf(pos);
}
static bool quantile_f32(const std::vector<float> &d, float q, float *out) {
    if (d.empty()) return false;
    size_t n = d.size();
    volatile float pos = q * (float)(n - 1);
    size_t idx = (size_t)floorf(pos);
    volatile float frac = pos - (float)idx;
    if (idx + 1 < n) {
        volatile float lo = d[idx] * (1.0f - frac);
        volatile float hi = d[idx + 1] * frac;
        *out = lo + hi;
    } else *out = d[idx];
    return true;
}

extern "C" int lrge_hip_median(const float *estimates, uint64_t n, int finite, int has_lower, float lower_q, int has_upper,
                               float upper_q, float out[3], int ok[3]) {
    if (!out || !ok || (n && !estimates)) return LRGE_ERR_INVALID;
    ok[0] = ok[1] = ok[2] = 0; out[0] = out[1] = out[2] = 0.f;
    if (!has_lower && has_upper) return LRGE_ERR_INVALID;  // the reference panics here (estimate.rs:109)
    if ((has_lower && !(lower_q >= 0.f && lower_q <= 1.f)) || (has_upper && !(upper_q >= 0.f && upper_q <= 1.f)))
        return LRGE_ERR_INVALID;                           // "Quantile must be between 0.0 and 1.0"
    std::vector<float> v;
    v.reserve(n);
    // kept values, and on the way a histogram over the top bits of their patterns: for non-negative floats the bit pattern
    // orders like the value, so the bin that holds an order statistic is known after one pass
    constexpr int kShift = 17, kBins = 1 << (31 - kShift);
    std::vector<u32> hist((size_t)kBins + 1, 0);
    bool radix_ok = true;
    v.resize(n);
    size_t nv = 0;
    for (u64 i = 0; i < n; ++i) {
        const float e = estimates[i];
        u32 b; memcpy(&b, &e, 4);
        if (finite && (b & 0x7F800000u) == 0x7F800000u) continue;      // infinity or NaN
        if ((b >> 31) || e != e) radix_ok = false; else ++hist[b >> kShift];
        v[nv++] = e;
    }
    v.resize(nv);
    if (v.empty()) return LRGE_OK;
    // the reference sorts the whole vector (estimate.rs:90-95); only the (at most six) order statistics the three
    // quantiles read are needed, and an order statistic does not depend on how ties are arranged
    std::vector<size_t> need;
    auto want = [&](float q) { const size_t i = quantile_index(v.size(), q); need.push_back(i); if (i + 1 < v.size()) need.push_back(i + 1); };
    want(0.5f);
    if (has_lower) want(lower_q);
    if (has_upper) want(upper_q);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    if (radix_ok) {
        // gather the (few) bins that hold a needed rank, select inside them, and put each statistic at its index of `v`
        // (quantile_f32 below reads v[idx] and v[idx + 1] only)
        std::vector<u32> cum((size_t)kBins + 1, 0);
        for (int b = 0; b < kBins; ++b) cum[(size_t)b + 1] = cum[b] + hist[b];
        std::vector<int> bin_of(need.size());
        std::vector<int> bins;
        for (size_t k = 0; k < need.size(); ++k) {
            const int b = (int)(std::upper_bound(cum.begin(), cum.end(), (u32)need[k]) - cum.begin()) - 1;
            bin_of[k] = b;
            if (bins.empty() || bins.back() != b) bins.push_back(b);       // (need is ascending, so are the bins)
        }
        std::vector<std::vector<float>> members(bins.size());
        for (size_t t = 0; t < bins.size(); ++t) members[t].reserve(hist[bins[t]]);
        std::vector<int8_t> slot_of((size_t)kBins, (int8_t)-1);      // (16 K bins: ~200 of 50 000 clustered estimates per bin)
        for (size_t t = 0; t < bins.size(); ++t) slot_of[bins[t]] = (int8_t)t;
        for (const float e : v) {
            u32 b; memcpy(&b, &e, 4);
            const int t = slot_of[b >> kShift];
            if (t >= 0) members[(size_t)t].push_back(e);
        }
        std::vector<float> stat(need.size());
        for (size_t k = 0; k < need.size(); ++k) {
            const size_t t = (size_t)(std::find(bins.begin(), bins.end(), bin_of[k]) - bins.begin());
            std::vector<float> &m = members[t];
            const size_t r = need[k] - cum[bin_of[k]];
            std::nth_element(m.begin(), m.begin() + r, m.end());
            stat[k] = m[r];
        }
        for (size_t k = 0; k < need.size(); ++k) v[need[k]] = stat[k];
    } else {
        // negative values or NaNs (finite == 0): comparison-based selection.  The middle one of the needed order statistics
        // first, then the rest inside the halves it leaves: every later selection works on a fraction of the vector
        struct Sel {
            static void run(std::vector<float> &v, const std::vector<size_t> &need, size_t a, size_t b, size_t lo, size_t hi) {
                if (a >= b) return;
                const size_t m = (a + b) / 2, i = need[m];
                std::nth_element(v.begin() + lo, v.begin() + i, v.begin() + hi);
                run(v, need, a, m, lo, i);
                run(v, need, m + 1, b, i + 1, hi);
            }
        };
        Sel::run(v, need, 0, need.size(), 0, v.size());
    }
    ok[1] = quantile_f32(v, 0.5f, &out[1]);
    if (has_lower) ok[0] = quantile_f32(v, lower_q, &out[0]);
    if (has_upper) ok[2] = quantile_f32(v, upper_q, &out[2]);
    return LRGE_OK;
}

extern "C" int lrge_hip_unique_random_set(uint64_t k, uint32_t n, int has_seed, uint64_t seed, uint32_t *out) {
    if (k > n || (k && !out)) return LRGE_ERR_INVALID;
    std::vector<uint32_t> v = lrge::unique_random_set((size_t)k, n, has_seed ? std::optional<uint64_t>(seed) : std::nullopt);
    std::copy(v.begin(), v.end(), out);
    return LRGE_OK;
}

extern "C" int lrge_hip_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    if (!key || !out || rounds <= 0 || (rounds & 1)) return LRGE_ERR_INVALID;
    lrge::rand09::chacha_block(key, counter, 0, rounds, out);
    return LRGE_OK;
}

// Host only (round 6): the records of an input file in any format liblrge accepts (io.rs:35-184; include/lrge_io.hpp, lrge_cram.hpp) --
// FASTA / FASTQ, SAM / unaligned BAM / unaligned CRAM 3.0, plain or gzip / bzip2 / xz / zstd -- through a callback(user, name, name
// length, bases, base count).  What non-C++ hosts use for the formats they do not parse themselves (lrge_amd/readio.py: CRAM).
// Errors (message in errbuf): LRGE_ERR_IO for an unreadable file, LRGE_ERR_PARSE for malformed input or a mapped record.
extern "C" int lrge_hip_read_records(const char *path, void (*cb)(void *, const char *, uint64_t, const char *, uint64_t), void *user,
                                     char *errbuf, uint64_t errcap) {
    if (!path || !cb) return LRGE_ERR_INVALID;
    auto fail = [&](int rc, const char *what) { if (errbuf && errcap) { snprintf(errbuf, (size_t)errcap, "%s", what); } g_last_error = what; return rc; };
    try {
        lrge::io::iter_records(path, [&](const std::string &n, const std::string &s) { cb(user, n.data(), (uint64_t)n.size(), s.data(), (uint64_t)s.size()); });
    } catch (const std::exception &e) { return fail(strncmp(e.what(), "cannot open", 11) == 0 ? LRGE_ERR_IO : LRGE_ERR_PARSE, e.what()); }
    return LRGE_OK;
}
