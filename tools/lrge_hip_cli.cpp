// lrge_hip_cli.cpp -- flag-for-flag mirror of the `lrge` command line (lrge/src/cli.rs:9-87,
// lrge/src/main.rs:32-123) driving the MI355X overlap engine through include/lrge_hip.hpp.
// Input: FASTA / FASTQ / unaligned SAM / unaligned BAM, plain or gzip / bzip2 / xz / zstd compressed, sniffed by
// magic bytes like liblrge/src/io.rs (include/lrge_io.hpp; CRAM is recognised and refused).  Like the reference CLI, -P is parsed but NOT forwarded to the
// builders (lrge/src/main.rs:56-85), so the preset is always ava-ont; pass --honour-platform to
// forward it (what the library API does).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <cmath>
#include <iostream>

#include "../include/lrge_hip.hpp"
#include "../include/lrge_io.hpp"

static lrge::Reads load_reads(const std::string &path) {   // io.rs:154-184 via include/lrge_io.hpp
    lrge::Reads r;
    try {
        lrge::io::iter_records(path, [&](const std::string &name, const std::string &seq) { r.names.push_back(name); r.seqs.push_back(seq); });
    } catch (const lrge::io::IoError &e) {
        throw lrge::LrgeError(LRGE_ERR_IO, e.what());
    }
    if (r.names.empty()) throw lrge::LrgeError(LRGE_ERR_IO, "IO error: Is the file empty?");   // count_records, io.rs:140-145
    return r;
}

// Rust's `{}` for an f32 (main.rs:107): the shortest decimal digits that read back as the same f32, written positionally
// (never with an exponent).
static std::string display_f32(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
    std::string m(buf);
    const size_t epos = m.find('e');
    const int ex = atoi(m.c_str() + epos + 1);
    std::string digits; bool neg = false;
    for (size_t i = 0; i < epos; ++i) { if (m[i] == '-') neg = true; else if (m[i] != '.') digits += m[i]; }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();      // 1.50e3 -> "15", exponent 3
    std::string out;
    if (ex >= 0) {
        if ((int)digits.size() <= ex + 1) out = digits + std::string((size_t)(ex + 1 - (int)digits.size()), '0');
        else out = digits.substr(0, (size_t)ex + 1) + "." + digits.substr((size_t)ex + 1);
    } else out = "0." + std::string((size_t)(-ex - 1), '0') + digits;
    return (neg ? "-" : "") + out;
}

int main(int argc, char **argv) {
    std::string input, output = "-", platform = "ont";
    std::optional<size_t> T = 10000, Q = 5000, N;
    bool T_set = false, Q_set = false, filter = false, with_inf = false, precise = false, use_min_ref = false, honour_platform = false;
    float q1 = lrge::LOWER_QUANTILE, q3 = lrge::UPPER_QUANTILE, ratio = 0.2f;
    size_t threads = 1; std::optional<uint64_t> seed; int quiet = 0, verbose = 0, device = 0;
    bool keep_temp = false, dump_records = false; std::string temp_dir;
    auto need = [&](int &i) -> const char * { if (i + 1 >= argc) { fprintf(stderr, "error: missing value for %s\n", argv[i]); exit(2); } return argv[++i]; };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-o" || a == "--output") output = need(i);
        else if (a == "-T" || a == "--target") { T = strtoull(need(i), 0, 10); T_set = true; }
        else if (a == "-Q" || a == "--query") { Q = strtoull(need(i), 0, 10); Q_set = true; }
        else if (a == "-n" || a == "--num") N = strtoull(need(i), 0, 10);
        else if (a == "-P" || a == "--platform") { platform = need(i); if (platform != "ont" && platform != "pb") { fprintf(stderr, "error: invalid platform\n"); return 2; } }
        else if (a == "-F" || a == "--filter-contained") filter = true;
        else if (a == "-t" || a == "--threads") threads = strtoull(need(i), 0, 10);
        else if (a == "-C" || a == "--keep-temp") keep_temp = true;
        else if (a == "-D" || a == "--temp") temp_dir = need(i);
        else if (a == "-s" || a == "--seed") seed = strtoull(need(i), 0, 10);
        else if (a == "-8" || a == "--inf") with_inf = true;
        else if (a == "-f" || a == "--float-my-boat") precise = true;
        else if (a == "--q1") q1 = strtof(need(i), 0);
        else if (a == "--q3") q3 = strtof(need(i), 0);
        else if (a == "--max-overhang-ratio") ratio = strtof(need(i), 0);
        else if (a == "--use-min-ref") use_min_ref = true;
        else if (a == "--honour-platform") honour_platform = true;
        else if (a == "--device") device = atoi(need(i));
        else if (a == "--dump-records") dump_records = true;   // host-only: print "id<TAB>sequence" per record and exit (tests)
        else if (a == "-q" || a == "--quiet") ++quiet; else if (a == "-qq") quiet += 2; else if (a == "-qqq") quiet += 3;
        else if (a == "-v" || a == "--verbose") ++verbose; else if (a == "-vv") verbose += 2;
        else if (!a.empty() && a[0] == '-' && a != "-") { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else input = a;
    }
    if (input.empty()) { fprintf(stderr, "error: the following required arguments were not provided: <INPUT>\n"); return 2; }
    if (N && (T_set || Q_set)) { fprintf(stderr, "error: the argument '--num <INT>' cannot be used with '--target'/'--query'\n"); return 2; }
    if (!(q1 >= 0.f && q1 <= 0.5f) || !(q3 >= 0.5f && q3 <= 1.f) || !(ratio >= 0.f && ratio <= 1.f)) { fprintf(stderr, "error: quantile/ratio out of range\n"); return 2; }
    if (quiet && verbose) { fprintf(stderr, "error: --quiet cannot be used with --verbose\n"); return 2; }
    const bool info = quiet == 0;
    try {
        if (dump_records) {
            size_t n = lrge::io::count_records(input);
            lrge::io::iter_records(input, [](const std::string &name, const std::string &seq) { printf("%s\t%s\n", name.c_str(), seq.c_str()); });
            fprintf(stderr, "%zu records\n", n);
            return 0;
        }
        lrge::Reads reads = load_reads(input);
        const lrge::Platform pf = (honour_platform && platform == "pb") ? lrge::Platform::PacBio : lrge::Platform::Nanopore;
        lrge::twoset::TwoSetStrategy ts(reads); lrge::ava::AvaStrategy as(reads);
        lrge::Estimate *st;
        if (N) {
            if (info) fprintf(stderr, "[INFO] Running all-vs-all strategy with %zu reads\n", *N);
            as = lrge::ava::Builder().num_reads(*N).remove_internal(filter, ratio).threads(threads).seed(seed).platform(pf).device(device).build(reads);
            st = &as;
        } else {
            if (info) fprintf(stderr, "[INFO] Running two-set strategy with %zu target reads and %zu query reads\n", *T, *Q);
            ts = lrge::twoset::Builder().target_num_reads(*T).query_num_reads(*Q).remove_internal(filter, ratio).use_min_ref(use_min_ref)
                     .threads(threads).seed(seed).platform(pf).device(device).build(reads);
            st = &ts;
        }
        std::vector<std::string> paf;
        if (keep_temp) { ts.paf_sink = &paf; as.paf_sink = &paf; }   // -C keeps overlaps.paf (main.rs:37-48, twoset.rs:246-250)
        lrge::EstimateResult r = st->estimate(!with_inf, q1, q3);
        if (keep_temp) {
            const std::string dir = temp_dir.empty() ? "." : temp_dir;
            std::ofstream pf(dir + "/overlaps.paf");
            if (!pf) { fprintf(stderr, "Error: Failed to write PAF record\n"); return 1; }
            for (auto &l : paf) pf << l << "\n";
            if (info) fprintf(stderr, "[INFO] Created temporary directory at %s\n", dir.c_str());
        }
        if (quiet < 2) for (auto &w : (N ? as.warnings : ts.warnings)) fprintf(stderr, "[WARN] %s\n", w.c_str());
        if (!r.estimate) { fprintf(stderr, "Error: %s\n", with_inf ? "No estimates were generated" : "No finite estimates were generated"); return 1; }
        if (info) {
            std::string msg = "Estimated genome size: " + lrge::format_estimate(*r.estimate);
            if (r.lower && r.upper) msg += " (IQR: " + lrge::format_estimate(*r.lower) + " - " + lrge::format_estimate(*r.upper) + ")";
            fprintf(stderr, "[INFO] %s\n", msg.c_str());
        }
        FILE *out = output == "-" ? stdout : fopen(output.c_str(), "w");
        if (!out) { fprintf(stderr, "Error: Failed to create output file\n"); return 1; }
        if (precise) fprintf(out, "%s\n", display_f32(*r.estimate).c_str()); else fprintf(out, "%.0f\n", (double)*r.estimate);
        if (out != stdout) fclose(out);
        if (info) fprintf(stderr, "[INFO] Done!\n");
    } catch (const lrge::io::IoError &e) {
        fprintf(stderr, "Error: Failed to generate estimate\n\nCaused by:\n    %s\n", e.what());
        return 1;
    } catch (const lrge::LrgeError &e) {
        fprintf(stderr, "Error: Failed to generate estimate\n\nCaused by:\n    %s\n", e.what());
        return 1;
    }
    return 0;
}
