// lrge_hip.hpp -- C++ host-side mirror of liblrge's operator interface on top of the C ABI
// (include/lrge_hip.h).  Header-only.  The reference is Rust (no Rust toolchain in this image), so the
// compiled-language host side is C++; names, defaults and error behaviour follow the reference:
//   trait Estimate / EstimateResult        liblrge/src/estimate.rs:8-78
//   twoset::Builder / TwoSetStrategy       liblrge/src/twoset/builder.rs:22-185, twoset.rs:75-201,587-606
//   ava::Builder / AvaStrategy             liblrge/src/ava/builder.rs:19-153, ava.rs:71-161,369-382
//   LrgeError                              liblrge/src/error.rs:6-33
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <vector>

#include "lrge_hip.h"
#include "lrge_rand.hpp"

namespace lrge {

struct LrgeError : std::runtime_error {
    int code;   // LRGE_ERR_*
    LrgeError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

constexpr float LOWER_QUANTILE = 0.15f, UPPER_QUANTILE = 0.65f;   // estimate.rs:40-46
enum class Platform { Nanopore, PacBio };                          // lib.rs:128-145

struct EstimateResult {   // estimate.rs:8-17
    std::optional<float> lower, estimate, upper;
    uint32_t no_mapping_count = 0;
};

struct Reads {            // parsed records: what io::iter_records hands to the strategies
    std::vector<std::string> names, seqs;
};

// trait Estimate (estimate.rs:21-78)
class Estimate {
public:
    virtual ~Estimate() = default;
    virtual std::pair<std::vector<float>, uint32_t> generate_estimates() = 0;
    EstimateResult estimate(bool finite, std::optional<float> lower_quant, std::optional<float> upper_quant) {
        auto [est, no_map] = generate_estimates();
        float out[3]; int ok[3];
        int rc = lrge_hip_median(est.data(), est.size(), finite ? 1 : 0, lower_quant ? 1 : 0, lower_quant.value_or(0.f),
                                 upper_quant ? 1 : 0, upper_quant.value_or(0.f), out, ok);
        if (rc) throw LrgeError(rc, "invalid quantile arguments");
        EstimateResult r;
        if (ok[0]) r.lower = out[0];
        if (ok[1]) r.estimate = out[1];
        if (ok[2]) r.upper = out[2];
        r.no_mapping_count = no_map;
        return r;
    }
};

namespace detail {
struct Ctx {
    lrge_hip_ctx *h = nullptr;
    explicit Ctx(int device) {
        int rc = lrge_hip_ctx_create(device, &h);
        if (rc) throw LrgeError(rc, std::string("device: ") + lrge_hip_last_error(nullptr));
    }
    ~Ctx() { lrge_hip_ctx_destroy(h); }
    void check(int rc) const { if (rc) throw LrgeError(rc, lrge_hip_last_error(h)); }
};
struct SeqSet {
    lrge_hip_seqset *h = nullptr;
    std::vector<uint32_t> lens;
    SeqSet(const Ctx &c, const std::vector<const std::string *> &seqs, const std::vector<uint32_t> &ranks) {
        std::vector<uint64_t> off(seqs.size() + 1, 0);
        for (size_t i = 0; i < seqs.size(); ++i) { off[i + 1] = off[i] + seqs[i]->size(); lens.push_back((uint32_t)seqs[i]->size()); }
        std::string cat; cat.reserve(off.back());
        for (auto *s : seqs) cat += *s;
        c.check(lrge_hip_seqset_upload(c.h, cat.data(), off.data(), (uint32_t)seqs.size(), ranks.data(), &h));
    }
    ~SeqSet() { lrge_hip_seqset_free(h); }
};
struct Index {
    lrge_hip_index *h = nullptr;
    Index(const Ctx &c, const SeqSet &s, int preset) { c.check(lrge_hip_index_build(c.h, s.h, preset, &h)); }
    ~Index() { lrge_hip_index_free(h); }
};
// strcmp ranks over the union of the given name lists (equal names share a rank)
inline std::vector<std::vector<uint32_t>> name_ranks(const std::vector<std::vector<const std::string *>> &lists) {
    std::vector<std::pair<const std::string *, std::pair<size_t, size_t>>> all;
    for (size_t l = 0; l < lists.size(); ++l) for (size_t i = 0; i < lists[l].size(); ++i) all.push_back({lists[l][i], {l, i}});
    std::sort(all.begin(), all.end(), [](auto &a, auto &b) { return *a.first < *b.first; });   // byte-wise, like strcmp
    std::vector<std::vector<uint32_t>> out(lists.size());
    for (size_t l = 0; l < lists.size(); ++l) out[l].resize(lists[l].size());
    uint32_t r = 0;
    for (size_t j = 0; j < all.size(); ++j) {
        if (j && *all[j].first != *all[j - 1].first) r = (uint32_t)j;
        out[all[j].second.first][all[j].second.second] = r;
    }
    return out;
}
// lib.rs:189-204: StdRng::seed_from_u64 + rand::seq::index::sample, restated in lrge_rand.hpp
using ::lrge::unique_random_set;
}  // namespace detail

inline std::vector<std::string> paf_lines(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *qs, int dual,
                                          const std::vector<const std::string *> &qn, const std::vector<uint32_t> &qlen,
                                          const std::vector<const std::string *> &tn, const std::vector<uint32_t> &tlen);

// ------------------------------------------------------------------------------------------
// two-set strategy
// ------------------------------------------------------------------------------------------
namespace twoset {
constexpr size_t DEFAULT_TARGET_NUM_READS = 10000, DEFAULT_QUERY_NUM_READS = 5000;   // twoset.rs:65-66

class TwoSetStrategy : public Estimate {
public:
    const Reads *input;
    size_t target_num_reads = DEFAULT_TARGET_NUM_READS, query_num_reads = DEFAULT_QUERY_NUM_READS;
    size_t target_num_bases = 0, query_num_bases = 0;
    bool remove_internal = false, use_min_ref = false;
    float max_overhang_ratio = 0.2f;
    size_t threads = 1;
    std::optional<uint64_t> seed;
    Platform platform = Platform::Nanopore;
    int device = 0;
    std::vector<std::string> warnings;
    std::vector<std::string> *paf_sink = nullptr;   // when set, receives the lines of overlaps.paf (the reference always writes it)

    explicit TwoSetStrategy(const Reads &r) : input(&r) {}

    // twoset.rs:122-201
    std::tuple<std::vector<size_t>, std::vector<size_t>, float> split_fastq() {
        const size_t n = input->names.size();
        if (n > 0xFFFFFFFFull) throw LrgeError(LRGE_ERR_TOO_MANY, "Number of reads in input file exceeds maximum allowed value");
        size_t n_req = target_num_reads + query_num_reads;
        if (n <= query_num_reads)
            throw LrgeError(LRGE_ERR_TOO_FEW, "Number of reads in input file (" + std::to_string(n) + ") is <= query number of reads (" +
                                                  std::to_string(query_num_reads) + ")");
        if (n < n_req) {
            warnings.push_back("Number of reads in input file (" + std::to_string(n) + ") is less than the sum of target and query reads (" +
                               std::to_string(n_req) + ")");
            target_num_reads = n - query_num_reads;
            n_req = n;
            warnings.push_back("Using " + std::to_string(target_num_reads) + " target reads");
        }
        auto idx = detail::unique_random_set(n_req, (uint32_t)n, seed);
        // split_into_hashsets (twoset.rs:632-652): the LAST target_num_reads sampled indices are the targets
        std::unordered_set<uint32_t> tset(idx.end() - (std::ptrdiff_t)target_num_reads, idx.end()),
            qset(idx.begin(), idx.end() - (std::ptrdiff_t)target_num_reads);
        std::vector<size_t> t, q;
        for (size_t i = 0; i < n; ++i) {   // file order, like iter_records
            if (tset.count((uint32_t)i)) { t.push_back(i); target_num_bases += input->seqs[i].size(); }
            else if (qset.count((uint32_t)i)) { q.push_back(i); query_num_bases += input->seqs[i].size(); }
        }
        const float avg_target_len = (float)target_num_bases / (float)target_num_reads;
        return {t, q, avg_target_len};
    }

    // twoset.rs:587-606 with align_reads / align_reads_inverse replaced by the device calls
    std::pair<std::vector<float>, uint32_t> generate_estimates() override {
        auto [t, q, avg_target_len] = split_fastq();
        std::vector<const std::string *> tn, ts, qn, qs;
        for (size_t i : t) { tn.push_back(&input->names[i]); ts.push_back(&input->seqs[i]); }
        for (size_t i : q) { qn.push_back(&input->names[i]); qs.push_back(&input->seqs[i]); }
        auto ranks = detail::name_ranks({qn, tn});
        detail::Ctx ctx(device);
        detail::SeqSet Q(ctx, qs, ranks[0]), T(ctx, ts, ranks[1]);
        const int preset = platform == Platform::PacBio ? LRGE_PRESET_AVA_PB : LRGE_PRESET_AVA_ONT;
        lrge_hip_params p{remove_internal ? 1 : 0, max_overhang_ratio};
        std::vector<uint32_t> counts(Q.lens.size()), has(Q.lens.size());
        uint32_t no_mapping = 0;
        if (use_min_ref && target_num_bases > query_num_bases) {
            ctx.check(lrge_hip_seqset_presketch(ctx.h, T.h, preset));   // streamed set: sketched beside the index build
            detail::Index ix(ctx, Q, preset);
            ctx.check(lrge_hip_overlap_inverse(ctx.h, ix.h, T.h, &p, counts.data()));
            if (paf_sink) *paf_sink = paf_lines(ctx.h, ix.h, T.h, 1, tn, T.lens, qn, Q.lens);
            for (uint32_t c : counts) no_mapping += c == 0;                       // twoset.rs:545-569
        } else {
            ctx.check(lrge_hip_seqset_presketch(ctx.h, Q.h, preset));
            detail::Index ix(ctx, T, preset);
            ctx.check(lrge_hip_overlap_twoset(ctx.h, ix.h, Q.h, &p, counts.data(), has.data()));
            if (paf_sink) *paf_sink = paf_lines(ctx.h, ix.h, Q.h, 1, qn, Q.lens, tn, T.lens);
            for (uint32_t h : has) no_mapping += h == 0;                          // twoset.rs:303-309
        }
        std::vector<float> est(counts.size());
        ctx.check(lrge_hip_estimates(ctx.h, counts.data(), Q.lens.data(), (uint32_t)counts.size(), avg_target_len,
                                     target_num_reads, 100, est.data()));
        return {est, no_mapping};
    }
};

class Builder {   // twoset/builder.rs:22-185
    size_t t_ = DEFAULT_TARGET_NUM_READS, q_ = DEFAULT_QUERY_NUM_READS, threads_ = 1;
    bool remove_internal_ = false, use_min_ref_ = false;
    float ratio_ = 0.2f;
    std::optional<uint64_t> seed_;
    Platform platform_ = Platform::Nanopore;
    int device_ = 0;
public:
    Builder &target_num_reads(size_t n) { t_ = n; return *this; }
    Builder &query_num_reads(size_t n) { q_ = n; return *this; }
    Builder &remove_internal(bool f, float ratio) { remove_internal_ = f; if (f) ratio_ = ratio; return *this; }
    Builder &use_min_ref(bool f) { use_min_ref_ = f; return *this; }
    Builder &threads(size_t n) { threads_ = n; return *this; }
    Builder &seed(std::optional<uint64_t> s) { seed_ = s; return *this; }
    Builder &platform(Platform p) { platform_ = p; return *this; }
    Builder &device(int d) { device_ = d; return *this; }
    TwoSetStrategy build(const Reads &input) const {
        TwoSetStrategy s(input);
        s.target_num_reads = t_; s.query_num_reads = q_; s.remove_internal = remove_internal_; s.max_overhang_ratio = ratio_;
        s.use_min_ref = use_min_ref_; s.threads = threads_; s.seed = seed_; s.platform = platform_; s.device = device_;
        return s;
    }
};
}  // namespace twoset

// ------------------------------------------------------------------------------------------
// all-vs-all strategy
// ------------------------------------------------------------------------------------------
namespace ava {
constexpr size_t DEFAULT_AVA_NUM_READS = 25000;   // ava.rs:62

class AvaStrategy : public Estimate {
public:
    const Reads *input;
    size_t num_reads = DEFAULT_AVA_NUM_READS, num_bases = 0, threads = 1;
    bool remove_internal = false;
    float max_overhang_ratio = 0.2f;
    std::optional<uint64_t> seed;
    Platform platform = Platform::Nanopore;
    int device = 0;
    std::vector<std::string> warnings;
    std::vector<std::string> *paf_sink = nullptr;

    explicit AvaStrategy(const Reads &r) : input(&r) {}

    std::pair<std::vector<float>, uint32_t> generate_estimates() override {
        // ava.rs:108-161
        const size_t n = input->names.size();
        if (n > 0xFFFFFFFFull) throw LrgeError(LRGE_ERR_TOO_MANY, "Number of reads in input file exceeds maximum allowed value");
        if (n < num_reads) {
            warnings.push_back("Number of reads in input file (" + std::to_string(n) + ") is less than the number requested (" +
                               std::to_string(num_reads) + ")");
            num_reads = n;
        }
        auto idx = detail::unique_random_set(num_reads, (uint32_t)n, seed);
        std::unordered_set<uint32_t> keep(idx.begin(), idx.end());
        std::vector<const std::string *> rn, rs;
        for (size_t i = 0; i < n; ++i) if (keep.count((uint32_t)i)) { rn.push_back(&input->names[i]); rs.push_back(&input->seqs[i]); num_bases += input->seqs[i].size(); }
        // ava.rs:369-382 + :165-366
        auto ranks = detail::name_ranks({rn});
        detail::Ctx ctx(device);
        detail::SeqSet R(ctx, rs, ranks[0]);
        const int preset = platform == Platform::PacBio ? LRGE_PRESET_AVA_PB : LRGE_PRESET_AVA_ONT;
        ctx.check(lrge_hip_seqset_presketch(ctx.h, R.h, preset));
        detail::Index ix(ctx, R, preset);
        lrge_hip_params p{remove_internal ? 1 : 0, max_overhang_ratio};
        std::vector<uint32_t> counts(R.lens.size());
        ctx.check(lrge_hip_overlap_ava(ctx.h, ix.h, R.h, &p, counts.data()));
        if (paf_sink) *paf_sink = paf_lines(ctx.h, ix.h, R.h, 0, rn, R.lens, rn, R.lens);
        const size_t n_target = num_reads - 1;                                   // ava.rs:339-346
        const float avg = (float)num_bases / (float)n_target;
        std::vector<float> est(counts.size());
        ctx.check(lrge_hip_estimates(ctx.h, counts.data(), R.lens.data(), (uint32_t)counts.size(), avg, n_target, 100, est.data()));
        uint32_t no_mapping = 0;
        for (uint32_t c : counts) no_mapping += c == 0;                          // ava.rs:329-331
        return {est, no_mapping};
    }
};

class Builder {   // ava/builder.rs:19-153
    size_t n_ = DEFAULT_AVA_NUM_READS, threads_ = 1;
    bool remove_internal_ = false;
    float ratio_ = 0.2f;
    std::optional<uint64_t> seed_;
    Platform platform_ = Platform::Nanopore;
    int device_ = 0;
public:
    Builder &num_reads(size_t n) { n_ = n; return *this; }
    Builder &remove_internal(bool f, float ratio) { remove_internal_ = f; if (f) ratio_ = ratio; return *this; }
    Builder &threads(size_t n) { threads_ = n; return *this; }
    Builder &seed(std::optional<uint64_t> s) { seed_ = s; return *this; }
    Builder &platform(Platform p) { platform_ = p; return *this; }
    Builder &device(int d) { device_ = d; return *this; }
    AvaStrategy build(const Reads &input) const {
        AvaStrategy s(input);
        s.num_reads = n_; s.remove_internal = remove_internal_; s.max_overhang_ratio = ratio_; s.threads = threads_;
        s.seed = seed_; s.platform = platform_; s.device = device_;
        return s;
    }
};
}  // namespace ava

// overlaps.paf (twoset.rs:246-250,289-293; mapping.rs:81-177): one line per chain, unordered like the reference
inline std::vector<std::string> paf_lines(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *qs, int dual,
                                          const std::vector<const std::string *> &qn, const std::vector<uint32_t> &qlen,
                                          const std::vector<const std::string *> &tn, const std::vector<uint32_t> &tlen) {
    uint64_t n = 0;
    int rc = lrge_hip_chains(ctx, ix, qs, dual, nullptr, 0, &n);
    if (rc) throw LrgeError(rc, lrge_hip_last_error(ctx));
    std::vector<lrge_hip_chain> ch(n ? n : 1);
    rc = lrge_hip_chains(ctx, ix, qs, dual, ch.data(), n, &n);
    if (rc) throw LrgeError(rc, lrge_hip_last_error(ctx));
    std::vector<int32_t> rl(qn.size() + 1); std::vector<uint64_t> ss(qn.size() + 1); std::vector<uint32_t> nk(qn.size() + 1);
    rc = lrge_hip_paf_stats(ctx, ix, qs, rl.data(), ss.data(), nk.data());
    if (rc) throw LrgeError(rc, lrge_hip_last_error(ctx));
    std::vector<std::string> out;
    char buf[512];
    for (uint64_t i = 0; i < n; ++i) {
        const lrge_hip_chain &c = ch[i];
        const uint32_t q = c.query, t = c.target;
        float dv = -1.0f;                                     // mm2:esterr.c mm_est_err
        if (nk[q]) {
            const float avg_k = (float)ss[q] / (float)nk[q];
            int n_tot = c.n_seeds;
            if ((float)c.qs > avg_k && (float)c.rs > avg_k) ++n_tot;
            if ((float)((int)qlen[q] - c.qs) > avg_k && (float)((int)tlen[t] - c.re) > avg_k) ++n_tot;
            dv = c.cnt >= n_tot ? 0.0f : (float)(1.0 - std::pow((double)c.cnt / n_tot, 1.0 / avg_k));
        }
        char dvs[32];
        if (dv < 1.1920929e-07f) snprintf(dvs, sizeof(dvs), "0"); else snprintf(dvs, sizeof(dvs), "%.4f", (double)dv);   // mapping.rs:136-147
        snprintf(buf, sizeof(buf), "\t%u\t%d\t%d\t%c\t", qlen[q], c.qs, c.qe, c.rev ? '-' : '+');
        std::string line = *qn[q] + buf + *tn[t];
        snprintf(buf, sizeof(buf), "\t%u\t%d\t%d\t%d\t%d\t0\ttp:A:S\tcm:i:%d\ts1:i:%d\tdv:f:%s\trl:i:%d", tlen[t], c.rs, c.re, c.mlen, c.blen,
                 c.cnt, c.score, dvs, rl[q]);
        out.push_back(line + buf);
    }
    return out;
}

// lrge/src/utils.rs:19-49
inline std::string format_estimate(float estimate) {
    if (std::isinf(estimate)) return "\xE2\x88\x9E bp";
    static const char *units[] = {"bp", "kbp", "Mbp", "Gbp", "Tbp", "Pbp"};
    float value = estimate; const char *suffix = "bp";
    for (int power = 0; power < 6; ++power) {
        const float threshold = std::pow(10.0f, (float)(power * 3));
        if (estimate >= threshold) { value = estimate / threshold; suffix = units[power]; } else break;
    }
    char buf[64];
    snprintf(buf, sizeof(buf), "%.2f %s", value, suffix);
    return buf;
}

}  // namespace lrge
