/*
 * lrge_hip.h -- C ABI of the MI355X-native overlap engine (liblrge_hip.so).
 *
 * Drop-in boundary for liblrge's overlap hot path.  The reference crosses its FFI seam once per
 * read (mm_map); a GPU wants batches, so each entry point below replaces a *group* of reference
 * calls.  All functions return 0 on success or a negative LRGE_ERR_* code; the message for the
 * last failure on a context is returned by lrge_hip_last_error().  Caller owns every input and
 * output buffer; the library owns the ctx / seqset / index handles (free with *_free/_destroy).
 * No callee-allocated memory crosses the ABI.  A ctx serves one call at a time.
 *
 * Reference interfaces replaced (paths under /root/reference/liblrge/src):
 *   lrge_hip_ctx_create        <- ThreadLocalBuffer / mm_tbuf_init (minimap2/thread_buf.rs:7-42)
 *   lrge_hip_seqset_upload     <- the (name, seq) records fed to Aligner::map / the FASTA read by
 *                                 mm_idx_reader_read (twoset.rs:216-241, minimap2/aligner.rs:171-185)
 *   lrge_hip_index_build       <- AlignerWrapper::new -> Aligner::builder/preset/dual/with_index
 *                                 (minimap2/aligner.rs:310-328, :53-122, :144-197): mm_set_opt,
 *                                 mm_idx_reader_open/read/close, mm_mapopt_update
 *   lrge_hip_overlap_twoset    <- TwoSetStrategy::align_reads (twoset.rs:204-367): per-read
 *                                 Aligner::map (aligner.rs:204-303) + distinct-target counting
 *   lrge_hip_overlap_inverse   <- TwoSetStrategy::align_reads_inverse (twoset.rs:370-584)
 *   lrge_hip_overlap_ava       <- AvaStrategy::align_reads (ava.rs:165-366)
 *   lrge_hip_estimates         <- estimate::per_read_estimate (estimate.rs:142-157)
 *   lrge_hip_median            <- estimate::median + calculate_quantile (estimate.rs:80-132)
 *   lrge_hip_unique_random_set <- unique_random_set (lib.rs:189-204): StdRng::seed_from_u64 +
 *                                 rand::seq::index::sample (rand 0.9.4 / rand_chacha 0.9.0, Cargo.lock:1001-1029)
 *   lrge_hip_chains            <- the Vec<PafRecord> of Aligner::map (aligner.rs:244-291,
 *                                 minimap2/mapping.rs:10-54), batched
 */
#ifndef LRGE_HIP_H
#define LRGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes map 1:1 onto LrgeError variants (error.rs:6-33) */
#define LRGE_OK                 0
#define LRGE_ERR_IO            -1  /* IoError */
#define LRGE_ERR_PARSE         -2  /* FastqParseError */
#define LRGE_ERR_TOO_MANY      -3  /* TooManyReadsError */
#define LRGE_ERR_TOO_FEW       -4  /* TooFewReadsError */
#define LRGE_ERR_DEVICE        -5  /* ThreadError class: HIP runtime / device failures */
#define LRGE_ERR_MAP           -6  /* MapError: "No index" / "Sequence is empty" (aligner.rs:210-216) */
#define LRGE_ERR_DUPLICATE_ID  -7  /* DuplicateReadIdentifier (ava.rs:195-199, twoset.rs:439-449) */
#define LRGE_ERR_PAF_WRITE     -8  /* PafWriteError */
#define LRGE_ERR_INVALID       -9  /* bad argument (no reference equivalent: would not compile in Rust) */

#define LRGE_PRESET_AVA_ONT 0   /* Preset::AvaOnt, "-k15 -Xw5 -e0 -m100 -r2k" (preset.rs:26) */
#define LRGE_PRESET_AVA_PB  1   /* Preset::AvaPb,  "-Hk19 -Xw5 -e0 -m100"     (preset.rs:24) */

typedef struct lrge_hip_ctx    lrge_hip_ctx;
typedef struct lrge_hip_seqset lrge_hip_seqset;
typedef struct lrge_hip_index  lrge_hip_index;
typedef struct lrge_hip_comm   lrge_hip_comm;

/* Builder knobs that reach the hot path (twoset/builder.rs:41-185, ava/builder.rs:38-153). */
typedef struct {
    int32_t remove_internal;     /* -F: drop overlaps the reference calls "internal" */
    float   max_overhang_ratio;  /* --max-overhang-ratio, default 0.2 (cli.rs:7) */
} lrge_hip_params;

/* One chain = one mm_reg1_t = one PafRecord (aligner.rs:253-290). */
typedef struct {
    uint32_t query;      /* index of the query read in its seqset */
    uint32_t target;     /* rid: index of the target read in the indexed seqset */
    int32_t  rev;        /* strand: 0 '+', 1 '-' */
    int32_t  score;      /* s1 */
    int32_t  cnt;        /* cm */
    int32_t  qs, qe;     /* query_start / query_end */
    int32_t  rs, re;     /* target_start / target_end */
    int32_t  mlen, blen; /* match_len / block_len */
    int32_t  n_seeds;    /* kept query seeds spanned by the chain: n_tot of mm_est_err before its two end
                            corrections; with cnt (= n_match) and the query's avg_k it gives dv */
} lrge_hip_chain;

/* Stage timings of the last overlap/index call on a ctx, in milliseconds (HIP event pairs recorded on the
   stream each stage runs on).  Index into the array with LRGE_T_*.  LRGE_T_CHAIN spans the whole chain stage
   (k_chain_hw beside k_chain_lpg, fork to join); LRGE_T_CHAIN_LPG is k_chain_lpg alone, on its side stream;
   LRGE_T_RS_SCATTER sums every k_rs_scatter launch of the call (they also count inside the sort stages). */
enum {
    LRGE_T_PACK = 0, LRGE_T_SKETCH, LRGE_T_INDEX_SORT, LRGE_T_INDEX_TABLE, LRGE_T_QFILTER,
    LRGE_T_LOOKUP, LRGE_T_EXPAND, LRGE_T_ANCHOR_SORT, LRGE_T_GROUP, LRGE_T_CHAIN,
    LRGE_T_CHAIN_GLB /* (unused: retired kernel) */, LRGE_T_COUNT, LRGE_T_TOTAL,
    LRGE_T_CHAIN_LPG, LRGE_T_RS_SCATTER, LRGE_T_K_LOOKUP /* k_lookup alone (it also counts inside LRGE_T_LOOKUP) */,
    LRGE_T_INDEX_RESTRICT /* lrge_hip_index_build_for: entry filter + global occurrence statistics */,
    LRGE_T_K_SKETCH /* the one-pass index sketch kernel alone (k_sketch_wave, or k_sketch_direct: see the two launch counters; they also count inside LRGE_T_SKETCH) */, LRGE_T_N
};
/* Work counters of the last overlap call (for the roofline's algorithmic bytes). */
enum {
    LRGE_C_QUERY_BASES = 0, LRGE_C_QUERY_MINIMIZERS, LRGE_C_ANCHORS, LRGE_C_GROUPS,
    LRGE_C_GROUPS_CHAINED, LRGE_C_CHAIN_LAUNCHES, LRGE_C_BATCHES,
    LRGE_C_CHAIN_ANCHORS /* anchors in chained groups */, LRGE_C_CHAIN_GLB_LAUNCHES, LRGE_C_CHAIN_GLB_ANCHORS,
    LRGE_C_LPG_LAUNCHES, LRGE_C_LPG_ANCHORS /* of those, anchors chained by k_chain_lpg */,
    LRGE_C_RS_SCATTER_LAUNCHES, LRGE_C_RS_SCATTER_ITEMS /* items moved by k_rs_scatter */,
    LRGE_C_RS_SCATTER_BYTES /* bytes those launches had to read + write: 32 per (key, value) pair, 16 per packed
                               key, 24 in the unpacking pass */,
    LRGE_C_LPG_SPLIT /* group size above which the last batch used k_chain_hw instead of k_chain_lpg */,
    LRGE_C_LOOKUP_LAUNCHES /* k_lookup launches (one per pass over a streamed set / index part) */,
    LRGE_C_TABLE_DISP_SUM /* last index build: sum over the distinct keys of their distance from the home slot in the ordered
                             table; divided by the number of keys it is ~0.5 at load 1/2 when the home slots are uniform */,
    LRGE_C_ANCHORS_KEPT /* of LRGE_C_ANCHORS (every seed hit expanded: minimap2's n_a), the anchors that left the expansion: the
                           dead-pair filter of count-only runs drops those of (target, strand) pairs too small to chain */,
    LRGE_C_INDEX_PARTS /* parts of the (partitioned) index the last overlap call went through, 0 = one index: LRGE_C_QUERY_MINIMIZERS and
                          LRGE_C_LOOKUP_LAUNCHES count every streamed minimizer once PER PART */,
    LRGE_C_SKETCH_LAUNCHES /* k_sketch_direct launches of the last index build / overlap call (LRGE_T_K_SKETCH) */,
    LRGE_C_SKETCH_WAVE_LAUNCHES /* k_sketch_wave launches (the wave-dense form of the index sketch; LRGE_T_K_SKETCH times whichever ran) */,
    LRGE_C_N
};

int  lrge_hip_device_count(int *n);
int  lrge_hip_ctx_create(int device, lrge_hip_ctx **out);
void lrge_hip_ctx_destroy(lrge_hip_ctx *ctx);
const char *lrge_hip_last_error(const lrge_hip_ctx *ctx);
/* Tuning / test options of a context (names and meanings: INTEGRATION.md section 6).  They are read from the environment
   exactly once, when the context is created (LRGE_HIP_<NAME>=value), and can be changed here afterwards; value NULL clears
   an option.  No entry point reads the environment per call.  The DEBUG_* options (fault injection, overrides of
   minimap2's chaining heuristics for the parity tests) are accepted through this call only, never from the environment. */
int  lrge_hip_ctx_set_option(lrge_hip_ctx *ctx, const char *name, const char *value);

/*
 * Upload a read set and pack it 2-bit (+ ambiguity mask) in HBM.
 *   bases     concatenated ASCII sequence bytes, offsets[n] bytes
 *   offsets   n+1 byte offsets into bases
 *   name_rank n lexicographic ranks of the read identifiers (header up to first whitespace,
 *             io.rs:199-204) computed over the UNION of all sets that will meet in one overlap
 *             call; equal names <=> equal rank.  Replaces strcmp(qname, tname) in minimap2's
 *             skip_seed and the name keys of liblrge's HashSet/HashMap.  NULL = all distinct.
 */
int  lrge_hip_seqset_upload(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets,
                            uint32_t n, const uint32_t *name_rank, lrge_hip_seqset **out);
/*
 * The same without waiting for the transfer: the copy and the 2-bit pack are queued on the context's copy stream and the
 * call returns; every later call that consumes the set orders itself behind them on the device.  So a second set travels
 * over PCIe while the first one is being indexed (the reference's producer thread / bounded channel, twoset.rs:216-241,
 * does the same for its per-read stream).  `bases` may be
 *   - pinned host memory (lrge_hip_host_alloc, hipHostMalloc/hipHostRegister): one DMA straight from the caller's buffer,
 *     which must stay valid and unchanged until a call that consumes the set has returned or lrge_hip_seqset_wait(s);
 *   - pageable host memory: staged through the context's pinned buffers (copied out before the call returns);
 *   - device memory (reads already resident in HBM as ASCII): packed in place, no transfer.
 * `offsets` and `name_rank` are copied before the call returns in every case.  lrge_hip_seqset_upload accepts the same
 * three kinds of `bases`.
 */
int  lrge_hip_seqset_upload_async(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets,
                                  uint32_t n, const uint32_t *name_rank, lrge_hip_seqset **out);
int  lrge_hip_seqset_wait(lrge_hip_seqset *s);          /* host-side wait for an async upload */
/* Pinned host memory for read buffers (what the record reader of io.rs:186-249 would fill): DMA source without staging. */
int  lrge_hip_host_alloc(size_t bytes, void **out);
void lrge_hip_host_free(void *p);
void lrge_hip_seqset_free(lrge_hip_seqset *s);
uint32_t lrge_hip_seqset_size(const lrge_hip_seqset *s);

/* Build the minimizer index over `targets` with the given preset (also fixes mid_occ). */
int  lrge_hip_index_build(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset,
                          lrge_hip_index **out);
/*
 * The index of a multi-GPU run (and optionally of a single GPU): built for ONE streamed set.  It holds the entries of the
 * minimizers that occur in `streamed` -- complete position lists, so mm_idx_get answers every question that set can ask
 * exactly as the full index would -- while mid_occ, n_keys and n_minimizers are those of the whole target set
 * (mm_idx_cal_max_occ over all distinct keys).  Only `streamed` (or the library's own views of it) may be streamed against
 * it: the other overlap calls fail with LRGE_ERR_INVALID.
 *   comm == NULL  one GPU: the occurrence statistics are counted here.
 *   comm != NULL  collective call: every rank passes the SAME target set and ITS OWN range of the streamed reads; each rank
 *                 counts a 1/world share of the hash space and one small all-reduce completes the histogram.  No index
 *                 data crosses the links.
 * What it replaces: AlignerWrapper::new (aligner.rs:310-328) called once per process in a run sharded by query
 * (twoset.rs:266-334).  streamed == NULL and comm == NULL is lrge_hip_index_build.
 */
int  lrge_hip_index_build_for(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset,
                              lrge_hip_seqset *streamed, lrge_hip_comm *comm, lrge_hip_index **out);
/*
 * The same index with the TARGET SKETCH sharded as well: a collective call in which every rank passes
 * ITS OWN contiguous share of the target reads -- reads [shard_first, shard_first + size(target_shard)) of the whole target
 * set, whose lengths and name ranks every rank knows (all_target_lens / all_target_ranks, n_targets entries; ranks may be
 * NULL) -- and its own range of the streamed reads.  Rank r sketches only its share; three exchanges complete the picture:
 * an all-gather of the ranks' key sets (Bloom filters), a variable-size all-to-all that sends every entry to the ranks whose
 * streamed reads carry its key (complete position lists wherever they are asked for), and one that sends every hash to the
 * rank that owns it for the occurrence statistics (one small all-reduce then fixes mid_occ, n_keys, n_minimizers globally).
 * The result answers the rank's streamed set exactly as the one index over all targets would.  The shares must be handed out
 * in rank order (rank 0 holds the first reads) so that position lists keep the order of the one index.
 * Replaces: AlignerWrapper::new (aligner.rs:310-328) in a run sharded by query (twoset.rs:266-334), with mm_idx_gen's
 * pipeline (mm2:index.c) itself spread over the ranks.
 */
int  lrge_hip_index_build_sharded(lrge_hip_ctx *ctx, const uint32_t *all_target_lens, const uint32_t *all_target_ranks,
                                  uint32_t n_targets, const lrge_hip_seqset *target_shard, uint32_t shard_first, int preset,
                                  lrge_hip_seqset *streamed, lrge_hip_comm *comm, lrge_hip_index **out);
/* The forward strategy over several GPUs with the TARGETS sharded (round 4; replaces, as the default multi-GPU form of
   twoset.rs:204-367, the query-sharded builds above -- those still stand).  Rank r indexes ITS contiguous share of the target
   reads (`target_shard`, uploaded on this context) and the caller maps ALL query reads against it with lrge_hip_overlap_twoset:
   the shards hold disjoint targets, so the per-query counts of the one index (aligner.rs:111-120) are the SUM of the ranks'
   counts (one lrge_hip_comm_allreduce_u32 of the count vector, and of has_mapping), bit for bit.  What is made global here is
   what mm_idx_cal_max_occ / mm_mapopt_update see (aligner.rs:189): every rank sends (hash, local count) per distinct key of its
   table to the hash's owner rank, the owners add up, one small all-reduce yields n_keys / n_minimizers / mid_occ of the whole
   target set (lrge_hip_index_stats reports those), and the keys above mid_occ are dropped in every rank's table.  No index entry
   crosses a link.  Collective: every rank of `comm` calls it; a failure on one rank fails it on all (the build ends with an
   agreement: no rank leaves with LRGE_OK alone).
   PRECONDITION the caller checks (it holds every target's name rank; lrge_amd/parallel.py: cross_shard_duplicates, the Rust shim):
   no target identifier occurs in two DIFFERENT shards.  The reference counts distinct target NAMES (twoset.rs:286-317) and never
   rejects a duplicate id in this mode; a duplicate inside one shard is counted once, one across shards would be counted once per
   shard.  (A partitioned single-GPU index has the same limit and checks it itself: LRGE_ERR_DUPLICATE_ID.) */
int  lrge_hip_index_build_tsharded(lrge_hip_ctx *ctx, const lrge_hip_seqset *target_shard, int preset, lrge_hip_comm *comm,
                                   lrge_hip_index **out);
/* Exchange volumes of the last lrge_hip_index_build_sharded on this context: {key-set bytes contributed, entries sketched here,
   entries sent to other ranks, entries received from other ranks, hashes sent, hashes received, bytes per entry | bytes per hash << 8
   (4 when the hash has at most 32 bits: k = 15), entries kept}. */
int  lrge_hip_last_shard_stats(const lrge_hip_ctx *ctx, uint64_t out[8]);
void lrge_hip_index_free(lrge_hip_index *ix);
int  lrge_hip_index_stats(const lrge_hip_index *ix, uint64_t *n_minimizers, uint64_t *n_keys,
                          int32_t *mid_occ);

/*
 * Two-set forward (dual = yes).  counts[q] = number of distinct target names among the kept
 * mappings of query q (twoset.rs:286-302); has_mapping[q] = 1 unless mappings.is_empty()
 * (twoset.rs:303-309).  Both arrays have lrge_hip_seqset_size(queries) entries.
 * A zero-length query is LRGE_ERR_MAP ("Sequence is empty").
 */
int  lrge_hip_overlap_twoset(lrge_hip_ctx *ctx, const lrge_hip_index *ix,
                             const lrge_hip_seqset *queries, const lrge_hip_params *p,
                             uint32_t *counts, uint32_t *has_mapping);
/*
 * Inverse two-set (--use-min-ref): `ix` indexes the QUERY set, `streamed` is the target set.
 * counts[i] (one per indexed read) += 1 for every streamed read with a kept mapping onto it
 * (twoset.rs:485-524).  Duplicate identifiers among the indexed reads: LRGE_ERR_DUPLICATE_ID.
 */
int  lrge_hip_overlap_inverse(lrge_hip_ctx *ctx, const lrge_hip_index *ix,
                              const lrge_hip_seqset *streamed, const lrge_hip_params *p,
                              uint32_t *counts);
/*
 * All-vs-all (dual = no): `reads` is the indexed set itself, or a SHARD of it (any subset, uploaded with name ranks
 * taken over the whole set; multi-GPU runs give every rank one shard).  counts[] has one entry per INDEXED read:
 * the symmetric overlap count (ava.rs:271-306) when `reads` is the whole set, this call's contribution to it
 * when it is a shard -- the sum over a partition of the reads is the all-vs-all result (NO_DUAL lets exactly one
 * read of every pair see it).  Duplicate identifiers: LRGE_ERR_DUPLICATE_ID.
 */
int  lrge_hip_overlap_ava(lrge_hip_ctx *ctx, const lrge_hip_index *ix,
                          const lrge_hip_seqset *reads, const lrge_hip_params *p,
                          uint32_t *counts);

/* Every chain of every query (the PafRecord stream), unordered.  *n_out receives the number of
   chains found; only the first `cap` are written.  dual: 1 = two-set flags, 0 = AVA flags. */
int  lrge_hip_chains(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries,
                     int dual, lrge_hip_chain *out, uint64_t cap, uint64_t *n_out);

/* Per-query seed statistics that complete the PafRecord tags (aligner.rs:262-270): rep_len = rl
   (query bases covered by seeds whose index occurrence exceeds mid_occ), and sum_span / n_kept, whose
   ratio (as f32) is mm_est_err's avg_k.  dv = n_match >= n_tot ? 0 : (float)(1.0 - pow((double)n_match /
   n_tot, 1.0 / avg_k)) with n_match = cnt and n_tot = n_seeds + [qs > avg_k && rs > avg_k] +
   [qlen - qs > avg_k && tlen - re > avg_k]  (mm2:esterr.c).  Arrays have one entry per query. */
int  lrge_hip_paf_stats(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries,
                        int32_t *rep_len, uint64_t *sum_span, uint32_t *n_kept);

/*
 * Multi-GPU: one context per GPU, every rank owns a range of the streamed reads end to end.  A
 * communicator carries the two exchanges of the path -- a SUM all-reduce of small integer vectors (the global
 * minimizer-occurrence histogram of lrge_hip_index_build_for; the per-indexed-read counts of the all-vs-all and inverse
 * strategies, ava.rs:300-301, twoset.rs:520-523) and an all-gather of the per-read estimates (the `estimates` vector of
 * twoset.rs:319-331).  Two transports:
 *   lrge_hip_comm_create        RCCL over xGMI, one PROCESS per GPU: rank 0 calls lrge_hip_comm_unique_id and hands the
 *                               128 bytes to the other ranks by whatever means the host has (a file, a socket, MPI,
 *                               torch.distributed's store); every rank then calls lrge_hip_comm_create collectively.
 *   lrge_hip_comm_create_local  the ranks are THREADS of one process (what a host that drives the GPUs from a thread
 *                               pool wants): buffers meet in host memory behind a barrier.  All ranks share one group
 *                               handle; collectives must be called by every rank, each from its own thread.
 * Collectives are blocking and must be entered by all ranks in the same order.  Host-buffer forms below; the library
 * itself uses the device forms inside lrge_hip_index_build_for.
 */
#define LRGE_HIP_COMM_ID_BYTES 128
int  lrge_hip_comm_unique_id(void *id128);
int  lrge_hip_comm_create(lrge_hip_ctx *ctx, int rank, int world, const void *id128, lrge_hip_comm **out);
int  lrge_hip_comm_local_group_create(int world, void **group);
void lrge_hip_comm_local_group_destroy(void *group);
int  lrge_hip_comm_create_local(lrge_hip_ctx *ctx, int rank, void *group, lrge_hip_comm **out);
/*   lrge_hip_comm_create_host   the host's own collectives (MPI, gloo, ...): the library stages its small vectors through host
 *                               memory and calls back.  allreduce: in-place SUM of n elements of elem_bytes (4: u32, 8: u64);
 *                               allgather: recv holds world * bytes; both return 0 on success. */
typedef int (*lrge_hip_host_allreduce_fn)(void *user, void *inout, size_t n, int elem_bytes);
typedef int (*lrge_hip_host_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
int  lrge_hip_comm_create_host(lrge_hip_ctx *ctx, int rank, int world, lrge_hip_host_allreduce_fn allreduce,
                               lrge_hip_host_allgather_fn allgather, void *user, lrge_hip_comm **out);
void lrge_hip_comm_destroy(lrge_hip_comm *c);
int  lrge_hip_comm_rank(const lrge_hip_comm *c);
int  lrge_hip_comm_world(const lrge_hip_comm *c);
int  lrge_hip_comm_allreduce_u32(lrge_hip_comm *c, uint32_t *inout, size_t n);                 /* in-place sum */
int  lrge_hip_comm_allgather(lrge_hip_comm *c, const void *send, size_t bytes, void *recv);    /* recv: world * bytes */
/* Variable-size all-to-all: this rank sends elements [send_off[d], send_off[d + 1]) of `send` to rank d and finds rank s's
   share at element recv_off[s] of `recv` (world + 1 prefix sums each, in elements of elem_bytes bytes; the receive counts are
   the other ranks' send counts -- exchange them with lrge_hip_comm_allgather first).  RCCL: one send / receive pair per
   peer in a group (point-to-point over xGMI); local groups copy device to device; host callbacks fall back to an all-gather.
   The exchange lrge_hip_index_build_sharded runs on device buffers. */
int  lrge_hip_comm_alltoallv(lrge_hip_comm *c, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off,
                             size_t elem_bytes);
/* Ranks RCCL itself counts in this communicator (ncclCommCount); 0 for the local / host transports. */
/* A rank that cannot go on between two collectives (a failed upload before a collective index build, a failed overlap call before
 * the all-reduce that closes the step) calls this instead of leaving its peers waiting for it: the reference's workers propagate a
 * MapError by ending the whole run (twoset.rs:279-284).  local: the group is aborted -- every rank blocked in, or later entering, one
 * of its collectives returns LRGE_ERR_DEVICE; RCCL: ncclCommAbort on this rank's communicator (its peers end with their own
 * processes: the launcher's job); host callbacks: the caller owns the collectives.  Afterwards every collective on c fails at once;
 * lrge_hip_comm_destroy is still to be called. */
int  lrge_hip_comm_abort(lrge_hip_comm *c);
int  lrge_hip_comm_rccl_ranks(const lrge_hip_comm *c, int *n);
/* librccl data-path calls (collectives, send / receive groups) made through c so far.  With LRGE_HIP_RCCL_WORLD1=1 (read when the RCCL
   communicator is created) a world of ONE goes through librccl for every collective instead of the world-1 shortcuts -- what lets a
   1-GPU box execute the RCCL branches with the shapes the sharded builds use (tests). */
int  lrge_hip_comm_rccl_ops(const lrge_hip_comm *c, uint64_t *n);
/* Timing emulation of a world on ONE GPU (local groups only): with serialize on, the ranks of the group take turns -- a rank
   computes between lrge_hip_comm_local_turn(c, 1) and (c, 0) and hands the GPU over whenever it waits for the others inside
   a collective; lrge_hip_comm_busy_ms returns the time it held the turn (what its share of the job takes on a GPU of its
   own, link transfers aside).  Results are unaffected.  on = 2: a rank that hands the GPU over also returns the idle segments of its
   device arena to the runtime (N arenas each sized for a whole GPU do not fit one), and the time inside the device allocator
   -- which a warm arena on a GPU of its own does not pay -- is kept out of busy_ms. */
int  lrge_hip_comm_local_group_serialize(void *group, int on);
int  lrge_hip_comm_local_turn(lrge_hip_comm *c, int begin);
double lrge_hip_comm_busy_ms(lrge_hip_comm *c, int reset);
/* (local groups) of that time, the part spent in the device-to-device copies that stand in for the link transfers of the
   variable-size all-gather (lrge_hip_seqset_presketch_sharded): on a node the links deliver into HBM and no copy is paid, so a
   projection takes busy - standin and adds the link model's time for the same bytes. */
double lrge_hip_comm_standin_ms(lrge_hip_comm *c, int reset);

/* per_read_estimate over n reads on the device (f32, no contraction). out[i] = +inf if counts[i]==0 */
int  lrge_hip_estimates(lrge_hip_ctx *ctx, const uint32_t *counts, const uint32_t *read_lens,
                        uint32_t n, float avg_target_len, uint64_t n_target_reads,
                        uint32_t overlap_thresh, float *out);
/* Host tail: median and optional quantiles of the (finite) estimates.  has_* mirror Option<f32>.
   out = {lower, median, upper}, ok[i] = 1 if that slot is Some. */
int  lrge_hip_median(const float *estimates, uint64_t n, int finite, int has_lower, float lower_q,
                     int has_upper, float upper_q, float out[3], int ok[3]);

/* Optional hint, results unchanged: sketch `s` ahead of the overlap call that will stream it.  The request is picked up
   by the NEXT lrge_hip_index_build on this context, which queues the set's sketch on a side stream right behind the
   index's own sketch, so that this VALU-bound work runs beside the index's memory-bound sort and table passes (mm_map
   sketches each query inside the call, mm2:map.c:mm_map_frag; here the set is known before the index exists).  The
   next lrge_hip_overlap_* call that streams `s` against an index of the same preset consumes the result (once); any
   other use simply sketches in line.  LRGE_HIP_NO_PRESKETCH=1 ignores the hint. */
int  lrge_hip_seqset_presketch(lrge_hip_ctx *ctx, lrge_hip_seqset *s, int preset);
/* The streamed set's sketch made ONCE per world instead of once per rank (round 6; the target-sharded forward strategy above has
   every rank map ALL queries, so every rank used to sketch all of them).  Collective: every rank of `comm` passes the SAME read set
   (same reads, same order: the ranks cut it by bases from the lengths they all hold); rank r runs mm_sketch (mm2:sketch.c, called per
   query by mm2:map.c collect_minimizers under Aligner::map, aligner.rs:231-241; twoset.rs:266-334 maps the queries independently of
   each other, so where a query is sketched is free) over its share only and the minimizers are all-gathered -- 16 bytes each, in read
   order, exactly the stream one rank's sketch of the whole set yields.  The result is attached to `s` like lrge_hip_seqset_presketch's
   and consumed (once) by the next lrge_hip_overlap_* call that streams `s` against an index of the same preset.  A failure on one rank
   fails the call on every rank (it ends with an agreement).  Sets of 2^32 bases or more: LRGE_ERR_TOO_MANY (they are streamed in views
   and sketched per view). */
int  lrge_hip_seqset_presketch_sharded(lrge_hip_ctx *ctx, lrge_hip_seqset *s, int preset, lrge_hip_comm *comm);

/* Host only: every record of an input file in any format liblrge accepts (io.rs:35-184: FASTA / FASTQ, SAM, unaligned BAM, unaligned
   CRAM 3.0; plain, gzip, bzip2, xz, zstd) through cb(user, name, name length, bases, base count) -- the C++ readers of
   include/lrge_io.hpp / lrge_cram.hpp behind a C entry point, for hosts that do not parse a format themselves.  A mapped record
   is refused with the reference's message (io.rs:162-167).  LRGE_ERR_IO: the file cannot be read; LRGE_ERR_PARSE: malformed
   input; the message goes to errbuf (may be NULL). */
int  lrge_hip_read_records(const char *path, void (*cb)(void *user, const char *name, uint64_t name_len, const char *bases, uint64_t n_bases),
                           void *user, char *errbuf, uint64_t errcap);

/* Host only: which side packs the reads of a set that starts in host memory when `ranks_on_host` ranks share this host's CPUs (option
   LRGE_HIP_RANKS_ON_HOST, set by the launcher; LRGE_HIP_PACK = host | device overrides): 1 = the host (2-bit pack with AVX2, packed words
   over PCIe), 0 = the device (ASCII over the rank's own PCIe link, k_pack).  *granted_cpus (may be NULL) receives the CPUs the host
   grants this process (affinity mask and cgroup bandwidth).  The rule: one rank -> host; several -> host only if every rank has 8 CPUs. */
int  lrge_hip_pack_choice(int ranks_on_host, double *granted_cpus);

/* Host only: the k distinct indices in [0, n) that liblrge's sub-sampling draws (lib.rs:189-204), in the order
   rand 0.9.4's index::sample returns them (split_into_hashsets, twoset.rs:632-652, takes the LAST target_num_reads
   of them as targets).  has_seed = 0 seeds the generator from OS entropy.  Restated in include/lrge_rand.hpp.
   LRGE_ERR_INVALID if k > n (the reference panics) or out is NULL with k > 0. */
int  lrge_hip_unique_random_set(uint64_t k, uint32_t n, int has_seed, uint64_t seed, uint32_t *out);
/* One ChaCha block of that generator (key = 8 LE words, 64-bit counter, stream 0): known-answer tests. */
int  lrge_hip_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]);

/* Stage-level introspection (parity tests, profiling). */
int  lrge_hip_sketch_dump(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, int preset, uint64_t *x,
                          uint64_t *y, uint64_t cap, uint64_t *n_out);
int  lrge_hip_index_dump(lrge_hip_ctx *ctx, const lrge_hip_index *ix, uint64_t *keys,
                         uint64_t *pos, uint64_t cap, uint64_t *n_out);
int  lrge_hip_anchors_dump(lrge_hip_ctx *ctx, const lrge_hip_index *ix,
                           const lrge_hip_seqset *queries, int dual, uint32_t query, uint64_t *x,
                           uint64_t *y, uint64_t cap, uint64_t *n_out);
/* How much of a call is timed with HIP events: 0 = LRGE_T_TOTAL, LRGE_T_CHAIN, LRGE_T_CHAIN_LPG only; 1 (default) = every
   stage; 2 = also a pair around every k_rs_scatter launch (LRGE_T_RS_SCATTER: ~14 more pairs per call, ~2 % of a C2
   step in host work between launches).  Untimed slots read 0.  LRGE_HIP_TIMERS=<level> sets the initial level. */
int  lrge_hip_set_timer_level(lrge_hip_ctx *ctx, int level);
int  lrge_hip_last_timings(const lrge_hip_ctx *ctx, float ms[LRGE_T_N]);
int  lrge_hip_last_counters(const lrge_hip_ctx *ctx, uint64_t c[LRGE_C_N]);
const char *lrge_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
