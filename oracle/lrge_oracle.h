/*
 * lrge_oracle.h -- CPU oracle for the liblrge overlap hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm the reference runs between "reads are in
 * memory" and "per-read genome-size estimates exist":
 *   liblrge (Rust)  liblrge/src/{twoset.rs,ava.rs,estimate.rs,minimap2/aligner.rs,minimap2/mapping.rs}
 *   minimap2 2.30   (crates.io minimap2-sys 0.1.30+minimap2.2.30, Cargo.lock:710-719 -- NOT present
 *                    under /root/reference; restated from its published algorithm: sketch.c, index.c,
 *                    seed.c, map.c, lchain.c, hit.c, esterr.c, options.c, ksort.h)
 *
 * PARITY STATUS: "parity unpinned" at the mm_map boundary -- the reference holds no golden vector
 * for overlap counts, and neither cargo/rustc nor a minimap2 binary/source exists in this image, so
 * the restatement could not be run against the real thing.  What IS pinned to the reference's own
 * known-answer tests: per_read_estimate (estimate.rs:305-342), median/quantiles
 * (estimate.rs:163-295), is_internal (mapping.rs:423-492).  See tests/test_oracle_kat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (lrge_amd/, include/lrge_hip.h) never links, imports or calls it.
 */
#ifndef LRGE_ORACLE_H
#define LRGE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } lo_mm128_t;

/* minimap2 flag bits used on this path (minimap.h) */
#define LO_F_NO_DIAG    0x001
#define LO_F_NO_DUAL    0x002
#define LO_F_NO_LJOIN   0x400
#define LO_F_ALL_CHAINS 0x800000

#define LO_PRESET_AVA_ONT 0
#define LO_PRESET_AVA_PB  1

/* tie-order policy of the three order-sensitive sorts (anchors by x, backtrack z by f) */
#define LO_SORT_STABLE 0   /* ties keep input order (what the HIP path implements)            */
#define LO_SORT_MM2    1   /* emulate ksort.h radix_sort_128x (MSD in-place + insertion sort) */

typedef struct {
    /* index options (mm_idxopt_t after mm_set_opt(0) + mm_set_opt(preset)) */
    int32_t k, w, is_hpc, bucket_bits;
    /* map options (mm_mapopt_t after preset, Aligner::dual, mm_mapopt_update) */
    int64_t flag;
    int32_t bw, bw_long, max_gap, max_gap_ref, max_chain_skip, max_chain_iter;
    int32_t min_cnt, min_chain_score, min_mid_occ, max_mid_occ, mid_occ, seed;
    float   mid_occ_frac, q_occ_frac, chain_gap_scale, chain_skip_scale;
    int32_t sort_mode; /* LO_SORT_* */
} lo_opt_t;

/* one chain == one mm_reg1_t == one PafRecord (aligner.rs:244-291) */
typedef struct {
    int32_t rid, rev, score, cnt;
    int32_t rs, re, qs, qe;
    int32_t mlen, blen;
    float   dv;
    int32_t rep_len;
} lo_reg_t;

typedef struct lo_index lo_index_t;

void lo_opt_init(lo_opt_t *o, int preset, int dual);
/* OpenMP threads of every parallel region that is not told otherwise */
void lo_set_default_threads(int n);

/* mm_sketch: returns number of minimizers (may exceed cap; only the first cap are written) */
int64_t lo_sketch(const char *seq, int32_t len, int32_t w, int32_t k, uint32_t rid, int32_t is_hpc,
                  lo_mm128_t *out, int64_t cap);
uint64_t lo_hash64(uint64_t key, uint64_t mask);

/* index over a read set: bases = concatenated ASCII, offs[n+1], names[n] NUL-terminated */
lo_index_t *lo_index_build(const char *bases, const uint64_t *offs, uint32_t n,
                           const char *const *names, lo_opt_t *opt /* mid_occ is filled in */);
void     lo_index_free(lo_index_t *ix);
int32_t  lo_index_mid_occ(const lo_index_t *ix);
uint64_t lo_index_n_minimizers(const lo_index_t *ix);
uint64_t lo_index_n_keys(const lo_index_t *ix);
/* mm_idx_get: number of hits for a minimizer hash (x>>8); *list points at the y values */
int32_t  lo_index_get(const lo_index_t *ix, uint64_t minier, const uint64_t **list);
/* the index is a SHARD of a larger target set: these keys are too frequent over the whole set (count > mid_occ there) and answer so
   here as well (tests of the target-sharded multi-GPU form); returns how many of them the shard holds */
uint64_t lo_index_drop_keys(lo_index_t *ix, const uint64_t *keys, uint64_t n);
/* one PART of a target set indexed part by part (see lrge_oracle.c): drop the sketch-order copy; list the keys whose local count is >=
   min_count (returns their number; the first cap are written); local counts of given keys */
void     lo_index_strip(lo_index_t *ix);
uint64_t lo_index_keys_at_least(const lo_index_t *ix, uint32_t min_count, uint64_t *out, uint64_t cap);
void     lo_index_counts_of(const lo_index_t *ix, const uint64_t *keys, uint64_t n, uint32_t *counts);
uint64_t lo_index_export_key_counts(const lo_index_t *ix, uint64_t *keys, uint8_t *counts);   /* all keys, counts saturated at 255 */
/* dump all minimizers in sketch order (rid-major) -- for stage-level parity tests */
uint64_t lo_index_dump_minimizers(const lo_index_t *ix, lo_mm128_t *out, uint64_t cap);

/* n_minimizers / n_keys / mid_occ of a target set sketched chunk by chunk (sets too large to index here in one piece);
   same arithmetic as lo_index_build's.  finish: 0, -1 out of memory, -2 the k-th count exceeds the histogram */
typedef struct lo_kstat lo_kstat_t;
lo_kstat_t *lo_kstat_new(const lo_opt_t *opt);
int  lo_kstat_add(lo_kstat_t *s, const char *bases, const uint64_t *offs, uint32_t n, int threads);
int  lo_kstat_finish(lo_kstat_t *s, int threads, uint64_t *n_mz, uint64_t *n_keys, int32_t *mid_occ);
void lo_kstat_free(lo_kstat_t *s);

/* The index of a target set too large to hold on the host, restricted to the keys a SAMPLE of query reads carries: complete
   position lists for exactly those keys (identical mm_idx_get answers for the sample), mid_occ handed in (lo_kstat_* gives
   it for the same set).  new: sketches the sample; add: the next n target reads, in rid order; finish: consumes the
   builder.  Usable with lo_map / lo_anchors / lo_twoset_counts for the sample queries only. */
typedef struct lo_ridx lo_ridx_t;
lo_ridx_t *lo_ridx_new(const lo_opt_t *opt, const char *qbases, const uint64_t *qoffs, uint32_t nq);
int  lo_ridx_add(lo_ridx_t *r, const char *bases, const uint64_t *offs, uint32_t n, const char *const *names, int threads);
lo_index_t *lo_ridx_finish(lo_ridx_t *r, lo_opt_t *opt, int32_t mid_occ, int threads);
void lo_ridx_free(lo_ridx_t *r);
uint64_t lo_ridx_n_sample_keys(const lo_ridx_t *r);
uint64_t lo_ridx_n_minimizers_seen(const lo_ridx_t *r);
uint64_t lo_ridx_n_kept(const lo_ridx_t *r);

/* stage outputs for one query: sorted anchors (after collect_seed_hits) */
int64_t lo_anchors(const lo_index_t *ix, const lo_opt_t *opt, const char *seq, int32_t qlen,
                   const char *qname, lo_mm128_t *out, int64_t cap);

/* mm_map: returns n_regs (may exceed cap) */
int32_t lo_map(const lo_index_t *ix, const lo_opt_t *opt, const char *seq, int32_t qlen,
               const char *qname, lo_reg_t *out, int32_t cap);

/* liblrge counting shells.  threads<=0 -> omp default */
int lo_twoset_counts(const lo_index_t *ix, const lo_opt_t *opt, const char *qbases,
                     const uint64_t *qoffs, uint32_t nq, const char *const *qnames,
                     int remove_internal, float max_overhang_ratio, int threads,
                     uint32_t *counts, uint32_t *has_mapping);
int lo_inverse_counts(const lo_index_t *ix /* index over QUERY set */, const lo_opt_t *opt,
                      const char *tbases, const uint64_t *toffs, uint32_t nt,
                      const char *const *tnames, int remove_internal, float max_overhang_ratio,
                      int threads, uint32_t *counts /* per indexed read */);
int lo_ava_counts(const lo_index_t *ix, const lo_opt_t *opt, const char *bases,
                  const uint64_t *offs, uint32_t n, const char *const *names, int remove_internal,
                  float max_overhang_ratio, int threads, uint32_t *counts);

/* estimate.rs */
float lo_per_read_estimate(uint64_t read_len, float avg_target_len, uint64_t n_target_reads,
                           uint64_t n_ovlaps, uint32_t ovlap_thresh);
/* median(): has_lower/has_upper mirror Option<f32>; out[3] = lower, median, upper; ok[3] flags.
   Returns 0, or -1 for the (None, Some) case the reference panics on (estimate.rs:109). */
int lo_median(const float *vals, uint64_t n, int finite_only, int has_lower, float lower_q,
              int has_upper, float upper_q, float out[3], int ok[3]);
/* mapping.rs:59-77 */
int lo_is_internal(int32_t qlen, int32_t qs, int32_t qe, int rev, int32_t tlen, int32_t ts,
                   int32_t te, float max_overhang_ratio);
/* twoset.rs:493-517 (the inverse-mode inline predicate: returns 1 when the mapping is SKIPPED) */
int lo_inverse_skip(int32_t qlen, int32_t qs, int32_t qe, int rev, int32_t tlen, int32_t ts,
                    int32_t te, float max_overhang_ratio);

/* the two tie policies as standalone sorts, for tests */
void lo_sort128x(lo_mm128_t *a, int64_t n, int mode);

#ifdef __cplusplus
}
#endif
#endif
