/*
 * lrge_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see lrge_oracle.h for the rules and the
 * "parity unpinned" statement).
 *
 * Each function cites what it restates.  Reference citations are into /root/reference
 * (liblrge/src/...).  "mm2:<file>:<function>" citations are into minimap2 v2.30, the third-party
 * dependency pinned by Cargo.lock:710-719 (minimap2-sys 0.1.30+minimap2.2.30) whose source is not
 * in the image; its published algorithm is restated here, anchored on the reference's call sites
 * (aligner.rs:56-63,74-80,171-192,231-241).
 */
#include "lrge_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* options: mm2:options.c:mm_idxopt_init, mm_mapopt_init, mm_set_opt("ava-ont"/"ava-pb");     */
/* call sites aligner.rs:53-87 (builder + preset), :93-103 (dual), :189 (mm_mapopt_update)     */
/* ------------------------------------------------------------------------------------------ */
void lo_opt_init(lo_opt_t *o, int preset, int dual)
{
    memset(o, 0, sizeof(*o));
    /* mm_idxopt_init */
    o->k = 15; o->w = 10; o->is_hpc = 0; o->bucket_bits = 14;
    /* mm_mapopt_init (fields that matter on this path) */
    o->seed = 11;
    o->mid_occ_frac = 2e-4f;
    o->min_mid_occ = 10;
    o->max_mid_occ = 1000000;
    o->q_occ_frac = 0.01f;
    o->min_cnt = 3;
    o->min_chain_score = 40;
    o->bw = 500; o->bw_long = 20000;
    o->max_gap = 5000; o->max_gap_ref = -1;
    o->max_chain_skip = 25; o->max_chain_iter = 5000;
    o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f;
    o->mid_occ = 0;
    /* presets: -k15 -Xw5 -e0 -m100 -r2k (ava-ont) / -Hk19 -Xw5 -e0 -m100 (ava-pb), preset.rs:24-26 */
    o->w = 5;
    o->flag |= LO_F_ALL_CHAINS | LO_F_NO_DIAG | LO_F_NO_DUAL | LO_F_NO_LJOIN;
    o->min_chain_score = 100;
    o->max_chain_skip = 25;
    if (preset == LO_PRESET_AVA_PB) {
        o->is_hpc = 1; o->k = 19;
        o->bw_long = o->bw;
    } else {
        o->is_hpc = 0; o->k = 15;
        o->bw = o->bw_long = 2000;
    }
    /* Aligner::dual (aligner.rs:93-103) */
    if (dual) o->flag &= ~(int64_t)LO_F_NO_DUAL; else o->flag |= LO_F_NO_DUAL;
    o->sort_mode = LO_SORT_STABLE;
}

/* ------------------------------------------------------------------------------------------ */
/* sketch: mm2:sketch.c:hash64, mm_sketch                                                      */
/* ------------------------------------------------------------------------------------------ */
static unsigned char nt4_of(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

uint64_t lo_hash64(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

typedef struct { lo_mm128_t *a; int64_t n, m; } vec128_t;

static void v128_push(vec128_t *v, lo_mm128_t e)
{
    if (v->n == v->m) {
        v->m = v->m ? v->m * 2 : 256;
        v->a = (lo_mm128_t *)realloc(v->a, (size_t)v->m * sizeof(lo_mm128_t));
    }
    v->a[v->n++] = e;
}

#define MAXU64 UINT64_MAX

static void sketch_into(const char *seq, int32_t len, int32_t w, int32_t k, uint32_t rid,
                        int32_t is_hpc, vec128_t *out)
{
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t kmer[2] = {0, 0};
    lo_mm128_t ring[256], cur_min = {MAXU64, MAXU64};
    int32_t hq[32], hq_front = 0, hq_count = 0; /* HPC run-length queue (last <=k runs) */
    int32_t i, j, l = 0, ring_pos = 0, min_pos = 0, kmer_span = 0;

    if (len <= 0 || w <= 0 || w >= 256 || k <= 0 || k > 28) return;
    for (j = 0; j < w; ++j) ring[j].x = ring[j].y = MAXU64;

    for (i = 0; i < len; ++i) {
        int c = nt4_of((unsigned char)seq[i]);
        lo_mm128_t info = {MAXU64, MAXU64};
        if (c < 4) {
            int z;
            if (is_hpc) {
                int32_t run = 1;
                if (i + 1 < len && nt4_of((unsigned char)seq[i + 1]) == c) {
                    for (run = 2; i + run < len; ++run)
                        if (nt4_of((unsigned char)seq[i + run]) != c) break;
                    i += run - 1; /* i now sits on the last base of the homopolymer run */
                }
                hq[(hq_count++ + hq_front) & 0x1f] = run;
                kmer_span += run;
                if (hq_count > k) { /* shift the oldest run out */
                    kmer_span -= hq[hq_front++];
                    hq_front &= 0x1f;
                    --hq_count;
                }
            } else kmer_span = l + 1 < k ? l + 1 : k;
            kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
            if (kmer[0] == kmer[1]) continue; /* strand-symmetric k-mer: skipped entirely */
            z = kmer[0] < kmer[1] ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = lo_hash64(kmer[z], mask) << 8 | (uint64_t)kmer_span;
                info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)z;
            }
        } else { l = 0; hq_count = hq_front = 0; kmer_span = 0; }
        ring[ring_pos] = info;
        if (l == w + k - 1 && cur_min.x != MAXU64) { /* first full window: flush equal minima */
            for (j = ring_pos + 1; j < w; ++j)
                if (cur_min.x == ring[j].x && ring[j].y != cur_min.y) v128_push(out, ring[j]);
            for (j = 0; j < ring_pos; ++j)
                if (cur_min.x == ring[j].x && ring[j].y != cur_min.y) v128_push(out, ring[j]);
        }
        if (info.x <= cur_min.x) { /* new minimum (right-most wins ties) */
            if (l >= w + k && cur_min.x != MAXU64) v128_push(out, cur_min);
            cur_min = info; min_pos = ring_pos;
        } else if (ring_pos == min_pos) { /* old minimum slid out of the window */
            if (l >= w + k - 1 && cur_min.x != MAXU64) v128_push(out, cur_min);
            cur_min.x = MAXU64;
            for (j = ring_pos + 1; j < w; ++j)
                if (cur_min.x >= ring[j].x) { cur_min = ring[j]; min_pos = j; }
            for (j = 0; j <= ring_pos; ++j)
                if (cur_min.x >= ring[j].x) { cur_min = ring[j]; min_pos = j; }
            if (l >= w + k - 1 && cur_min.x != MAXU64) {
                for (j = ring_pos + 1; j < w; ++j)
                    if (cur_min.x == ring[j].x && cur_min.y != ring[j].y) v128_push(out, ring[j]);
                for (j = 0; j <= ring_pos; ++j)
                    if (cur_min.x == ring[j].x && cur_min.y != ring[j].y) v128_push(out, ring[j]);
            }
        }
        if (++ring_pos == w) ring_pos = 0;
    }
    if (cur_min.x != MAXU64) v128_push(out, cur_min);
}

int64_t lo_sketch(const char *seq, int32_t len, int32_t w, int32_t k, uint32_t rid, int32_t is_hpc,
                  lo_mm128_t *out, int64_t cap)
{
    vec128_t v = {0, 0, 0};
    int64_t n;
    sketch_into(seq, len, w, k, rid, is_hpc, &v);
    n = v.n;
    if (out) memcpy(out, v.a, (size_t)(n < cap ? n : cap) * sizeof(lo_mm128_t));
    free(v.a);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* sorts: mm2:ksort.h:radix_sort_128x (LO_SORT_MM2) and a stable merge sort (LO_SORT_STABLE)  */
/* ------------------------------------------------------------------------------------------ */
#define MM2_RS_MIN 64

static void mm2_insertion(lo_mm128_t *beg, lo_mm128_t *end)
{
    lo_mm128_t *i;
    for (i = beg + 1; i < end; ++i)
        if (i->x < (i - 1)->x) {
            lo_mm128_t *j, tmp = *i;
            for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
            *j = tmp;
        }
}

typedef struct { lo_mm128_t *b, *e; } rs_bucket_t;

static void mm2_msd_pass(lo_mm128_t *beg, lo_mm128_t *end, int shift)
{
    rs_bucket_t bk[256], *k, *be = bk + 256;
    lo_mm128_t *i;
    for (k = bk; k != be; ++k) k->b = k->e = beg;
    for (i = beg; i != end; ++i) ++bk[i->x >> shift & 255].e;
    for (k = bk + 1; k != be; ++k) { k->e += (k - 1)->e - beg; k->b = (k - 1)->e; }
    for (k = bk; k != be;) { /* in-place cycle-leader permutation (not stable) */
        if (k->b != k->e) {
            rs_bucket_t *l = bk + (k->b->x >> shift & 255);
            if (l != k) {
                lo_mm128_t tmp = *k->b, swap;
                do {
                    swap = tmp; tmp = *l->b; *l->b++ = swap;
                    l = bk + (tmp.x >> shift & 255);
                } while (l != k);
                *k->b++ = tmp;
            } else ++k->b;
        } else ++k;
    }
    bk[0].b = beg;
    for (k = bk + 1; k != be; ++k) k->b = (k - 1)->e;
    if (shift) {
        shift = shift > 8 ? shift - 8 : 0;
        for (k = bk; k != be; ++k) {
            if (k->e - k->b > MM2_RS_MIN) mm2_msd_pass(k->b, k->e, shift);
            else if (k->e - k->b > 1) mm2_insertion(k->b, k->e);
        }
    }
}

static void stable_sort128x(lo_mm128_t *a, int64_t n)
{
    lo_mm128_t *tmp, *src, *dst;
    int64_t width, i;
    if (n < 2) return;
    tmp = (lo_mm128_t *)malloc((size_t)n * sizeof(lo_mm128_t));
    src = a; dst = tmp;
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int64_t l = i, m = i + width < n ? i + width : n, r = i + 2 * width < n ? i + 2 * width : n;
            int64_t p = l, q = m, o = l;
            while (p < m && q < r) dst[o++] = src[q].x < src[p].x ? src[q++] : src[p++];
            while (p < m) dst[o++] = src[p++];
            while (q < r) dst[o++] = src[q++];
        }
        { lo_mm128_t *t = src; src = dst; dst = t; }
    }
    if (src != a) memcpy(a, src, (size_t)n * sizeof(lo_mm128_t));
    free(tmp);
}

void lo_sort128x(lo_mm128_t *a, int64_t n, int mode)
{
    if (mode == LO_SORT_MM2) {
        if (n <= MM2_RS_MIN) mm2_insertion(a, a + n);
        else mm2_msd_pass(a, a + n, 56);
    } else stable_sort128x(a, n);
}

/* ------------------------------------------------------------------------------------------ */
/* index: mm2:index.c:mm_idx_gen (rid = file order), worker_post (group by x>>8, position    */
/* lists sorted by y), mm_idx_get, mm_idx_cal_max_occ; mm2:options.c:mm_mapopt_update         */
/* call site aligner.rs:171-192                                                               */
/* ------------------------------------------------------------------------------------------ */
struct lo_index {
    uint32_t n_seq;
    char **name;
    int32_t *len;
    uint32_t *name_rank;   /* lexicographic rank, equal names share a rank */
    int32_t k, w, is_hpc;
    uint64_t n_mz;         /* total minimizers */
    lo_mm128_t *mz;        /* sketch order (rid-major) */
    uint64_t n_keys;
    uint64_t *key;         /* sorted distinct x>>8 */
    uint64_t *off;         /* n_keys+1 offsets into pos[] */
    uint64_t *pos;         /* y values, ascending within a key */
    int32_t mid_occ;
    unsigned char *dropped; /* lo_index_drop_keys: keys that are too frequent over a LARGER target set this index is a shard of */
};

typedef struct { uint64_t h, y; } hy_t;

static int cmp_hy(const void *pa, const void *pb)
{
    const hy_t *a = (const hy_t *)pa, *b = (const hy_t *)pb;
    if (a->h != b->h) return a->h < b->h ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    return 0;
}

static int cmp_u32(const void *pa, const void *pb)
{
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    return a < b ? -1 : a > b;
}

typedef struct { const char *s; uint32_t i; } nameidx_t;
static int cmp_name(const void *pa, const void *pb)
{
    return strcmp(((const nameidx_t *)pa)->s, ((const nameidx_t *)pb)->s);
}

/* mm_idx_cal_max_occ(f) followed by the clamps of mm_mapopt_update */
static int32_t calc_mid_occ(const lo_index_t *ix, const lo_opt_t *o)
{
    int32_t thres;
    if (o->mid_occ_frac <= 0.f || ix->n_keys == 0) thres = INT32_MAX;
    else {
        uint64_t i, n = ix->n_keys;
        /* the k-th smallest count (mm2: ks_ksmall_uint32_t over all counts).  Small sets: sort, as before; large ones (a full
           H. sapiens-scale index has 4 x 10^8 keys): select through a histogram of the counts below 2^20 -- the same order statistic */
        uint64_t kth = (uint64_t)((1. - (double)o->mid_occ_frac) * (double)n); /* 0-based k-th smallest */
        if (n < (1u << 22)) {
            uint32_t *cnt = (uint32_t *)malloc(n * 4);
            for (i = 0; i < n; ++i) cnt[i] = (uint32_t)(ix->off[i + 1] - ix->off[i]);
            qsort(cnt, n, 4, cmp_u32);
            thres = (int32_t)(cnt[kth] + 1);
            free(cnt);
        } else {
            const uint32_t HB = 1u << 20;
            uint64_t *hist = (uint64_t *)calloc((size_t)HB + 1, 8), run = 0, c;
            uint32_t b, over_n = 0, *over = 0;
            for (i = 0; i < n; ++i) { c = ix->off[i + 1] - ix->off[i]; ++hist[c < HB ? c : HB]; }
            for (b = 0; b < HB && run + hist[b] <= kth; ++b) run += hist[b];
            if (b < HB) thres = (int32_t)(b + 1);
            else {      /* the k-th count is one of the few >= 2^20: sort those */
                over = (uint32_t *)malloc((hist[HB] ? hist[HB] : 1) * 4);
                for (i = 0; i < n; ++i) { c = ix->off[i + 1] - ix->off[i]; if (c >= HB) over[over_n++] = (uint32_t)c; }
                qsort(over, over_n, 4, cmp_u32);
                thres = (int32_t)(over[kth - run] + 1);
                free(over);
            }
            free(hist);
        }
    }
    if (thres < o->min_mid_occ) thres = o->min_mid_occ;
    if (o->max_mid_occ > o->min_mid_occ && thres > o->max_mid_occ) thres = o->max_mid_occ;
    return thres;
}

lo_index_t *lo_index_build(const char *bases, const uint64_t *offs, uint32_t n,
                           const char *const *names, lo_opt_t *opt)
{
    lo_index_t *ix = (lo_index_t *)calloc(1, sizeof(*ix));
    vec128_t v = {0, 0, 0};
    uint64_t i, j;
    hy_t *hy;
    nameidx_t *ni;

    ix->n_seq = n; ix->k = opt->k; ix->w = opt->w; ix->is_hpc = opt->is_hpc;
    ix->name = (char **)calloc(n ? n : 1, sizeof(char *));
    ix->len = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    ix->name_rank = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    ni = (nameidx_t *)malloc((n ? n : 1) * sizeof(nameidx_t));
    for (i = 0; i < n; ++i) {
        const char *nm = names && names[i] ? names[i] : "";
        ix->name[i] = (char *)malloc(strlen(nm) + 1);
        strcpy(ix->name[i], nm);
        ix->len[i] = (int32_t)(offs[i + 1] - offs[i]);
        ni[i].s = ix->name[i]; ni[i].i = (uint32_t)i;
    }
    {   /* sketch every read (minimap2 does this with its index threads: aligner.rs:181-185); reads are
           sketched independently and concatenated in rid order, so the result is thread-count independent */
        vec128_t *per = (vec128_t *)calloc(n ? n : 1, sizeof(vec128_t));
        int64_t r;
        uint64_t tot = 0;
#pragma omp parallel for schedule(dynamic, 16)
        for (r = 0; r < (int64_t)n; ++r)
            if (ix->len[r] > 0) /* zero-length targets keep a rid but are not sketched */
                sketch_into(bases + offs[r], ix->len[r], opt->w, opt->k, (uint32_t)r, opt->is_hpc, &per[r]);
        for (i = 0; i < n; ++i) tot += (uint64_t)per[i].n;
        v.a = (lo_mm128_t *)malloc((tot ? tot : 1) * sizeof(lo_mm128_t)); v.n = v.m = (int64_t)tot;
        for (i = 0, tot = 0; i < n; ++i) {
            if (per[i].n) memcpy(v.a + tot, per[i].a, (size_t)per[i].n * sizeof(lo_mm128_t));
            tot += (uint64_t)per[i].n; free(per[i].a);
        }
        free(per);
    }
    qsort(ni, n, sizeof(nameidx_t), cmp_name);
    for (i = 0, j = 0; i < n; ++i) {
        if (i > 0 && strcmp(ni[i].s, ni[i - 1].s) != 0) j = i;
        ix->name_rank[ni[i].i] = (uint32_t)j;
    }
    free(ni);

    ix->n_mz = (uint64_t)v.n; ix->mz = v.a;
    hy = (hy_t *)malloc((ix->n_mz ? ix->n_mz : 1) * sizeof(hy_t));
    {   /* sort by (hash, y): bucket on the top 12 hash bits (buckets concatenate into the global order),
           then sort the buckets in parallel.  The scatter is parallel too: (hash, y) is a total order on the entries of
           one index (y = rid << 32 | pos << 1 | strand is unique), so where an entry lands inside its bucket before the
           sort does not matter */
        const int shift = 2 * opt->k > 12 ? 2 * opt->k - 12 : 0;
        uint64_t *bstart = (uint64_t *)calloc(4097 + 1, 8), *loc;
        int64_t b;
        int nth = 1, t;
#pragma omp parallel
        {
#pragma omp single
            nth = omp_get_num_threads();
        }
        /* thread t scatters the t-th contiguous slice of the entries to positions of its own inside every bucket
           (counts per (thread, bucket), then one scan in bucket-major order): no atomics, no shared counters */
        loc = (uint64_t *)calloc((size_t)nth * 4096, 8);
        /* (slices are dealt by a loop over 0..nth, not by omp_get_thread_num(): whatever number of threads the runtime
           grants for a region -- OMP_DYNAMIC, OMP_THREAD_LIMIT, nesting -- every slice is counted and scattered) */
#pragma omp parallel for schedule(static, 1)
        for (t = 0; t < nth; ++t) {
            const int me = t;
            const uint64_t per = (ix->n_mz + (uint64_t)nth - 1) / (uint64_t)nth, lo = per * (uint64_t)me < ix->n_mz ? per * (uint64_t)me : ix->n_mz, hi = lo + per < ix->n_mz ? lo + per : ix->n_mz;
            uint64_t *l = loc + (size_t)me * 4096, q;
            for (q = lo; q < hi; ++q) ++l[(v.a[q].x >> 8) >> shift];
        }
        {
            uint64_t run = 0, bq;
            for (bq = 0; bq < 4096; ++bq) {
                bstart[bq] = run;
                for (t = 0; t < nth; ++t) { uint64_t c = loc[(size_t)t * 4096 + bq]; loc[(size_t)t * 4096 + bq] = run; run += c; }
            }
            bstart[4096] = run;
        }
#pragma omp parallel for schedule(static, 1)
        for (t = 0; t < nth; ++t) {
            const int me = t;
            const uint64_t per = (ix->n_mz + (uint64_t)nth - 1) / (uint64_t)nth, lo = per * (uint64_t)me < ix->n_mz ? per * (uint64_t)me : ix->n_mz, hi = lo + per < ix->n_mz ? lo + per : ix->n_mz;
            uint64_t *l = loc + (size_t)me * 4096, q;
            for (q = lo; q < hi; ++q) {
                const uint64_t h = v.a[q].x >> 8, d = l[h >> shift]++;
                hy[d].h = h; hy[d].y = v.a[q].y;
            }
        }
        free(loc);
#pragma omp parallel for schedule(dynamic, 8)
        for (b = 0; b < 4096; ++b)
            if (bstart[b + 1] > bstart[b]) qsort(hy + bstart[b], bstart[b + 1] - bstart[b], sizeof(hy_t), cmp_hy);
        free(bstart);
    }
    {   /* run heads -> key[], off[], pos[]: the heads are counted per block, the blocks' counts scanned, the arrays filled in parallel */
        const int64_t nblk = 1024;
        const uint64_t per = (ix->n_mz + (uint64_t)nblk - 1) / (uint64_t)nblk;
        uint64_t *bh = (uint64_t *)calloc((size_t)nblk + 1, 8);
        int64_t bb;
#pragma omp parallel for schedule(static)
        for (bb = 0; bb < nblk; ++bb) {
            uint64_t lo = (uint64_t)bb * per, hi = lo + per < ix->n_mz ? lo + per : ix->n_mz, t, c = 0;
            for (t = lo; t < hi; ++t) c += (t == 0 || hy[t].h != hy[t - 1].h);
            bh[bb + 1] = c;
        }
        for (bb = 0; bb < nblk; ++bb) bh[bb + 1] += bh[bb];
        j = bh[nblk];
        ix->n_keys = j;
        ix->key = (uint64_t *)malloc((j ? j : 1) * 8);
        ix->off = (uint64_t *)malloc((j + 1) * 8);
        ix->pos = (uint64_t *)malloc((ix->n_mz ? ix->n_mz : 1) * 8);
#pragma omp parallel for schedule(static)
        for (bb = 0; bb < nblk; ++bb) {
            uint64_t lo = (uint64_t)bb * per, hi = lo + per < ix->n_mz ? lo + per : ix->n_mz, t, jj = bh[bb];
            for (t = lo; t < hi; ++t) {
                if (t == 0 || hy[t].h != hy[t - 1].h) { ix->key[jj] = hy[t].h; ix->off[jj] = t; ++jj; }
                ix->pos[t] = hy[t].y;
            }
        }
        ix->off[j] = ix->n_mz;
        free(bh);
    }
    free(hy);

    /* mm_mapopt_update: only when mid_occ was not given */
    if (opt->mid_occ <= 0) opt->mid_occ = calc_mid_occ(ix, opt);
    if (opt->bw_long < opt->bw) opt->bw_long = opt->bw;
    ix->mid_occ = opt->mid_occ;
    return ix;
}

void lo_index_free(lo_index_t *ix)
{
    uint32_t i;
    if (!ix) return;
    for (i = 0; i < ix->n_seq; ++i) free(ix->name[i]);
    free(ix->name); free(ix->len); free(ix->name_rank); free(ix->mz);
    free(ix->key); free(ix->off); free(ix->pos); free(ix->dropped); free(ix);
}

int32_t  lo_index_mid_occ(const lo_index_t *ix) { return ix->mid_occ; }
uint64_t lo_index_n_minimizers(const lo_index_t *ix) { return ix->n_mz; }
uint64_t lo_index_n_keys(const lo_index_t *ix) { return ix->n_keys; }

/* A PART of a target set too large to index on this host in one piece (tools/c5_allcounts.py: H. sapiens scale in 8 parts, the
   target-sharded argument of tests/test_host_mirror.py with the oracle on every side): the sketch-order copy is dropped (only the
   minimizer dump reads it), and the keys whose LOCAL count reaches min_count are listed -- a key that is too frequent over all P parts
   (count > mid_occ) reaches ceil((mid_occ + 1) / P) in at least one of them, so the union of the parts' lists holds every candidate. */
void lo_index_strip(lo_index_t *ix) { if (ix) { free(ix->mz); ix->mz = 0; } }
uint64_t lo_index_keys_at_least(const lo_index_t *ix, uint32_t min_count, uint64_t *out, uint64_t cap)
{
    uint64_t i, n = 0;
    for (i = 0; i < ix->n_keys; ++i)
        if (ix->off[i + 1] - ix->off[i] >= min_count) { if (out && n < cap) out[n] = ix->key[i]; ++n; }
    return n;
}
/* every distinct key with its local count (saturated at 255), ascending: what is kept of a part whose index is dropped again
   (tools/c5_allcounts.py --two-pass); returns n_keys */
uint64_t lo_index_export_key_counts(const lo_index_t *ix, uint64_t *keys, uint8_t *counts)
{
    int64_t t;
#pragma omp parallel for schedule(static)
    for (t = 0; t < (int64_t)ix->n_keys; ++t) {
        const uint64_t c = ix->off[t + 1] - ix->off[t];
        if (keys) keys[t] = ix->key[t];
        if (counts) counts[t] = (uint8_t)(c < 255 ? c : 255);
    }
    return ix->n_keys;
}
/* local occurrence counts of n keys (0: absent), whatever lo_index_drop_keys has marked */
void lo_index_counts_of(const lo_index_t *ix, const uint64_t *keys, uint64_t n, uint32_t *counts)
{
    int64_t t;
#pragma omp parallel for schedule(static)
    for (t = 0; t < (int64_t)n; ++t) {
        uint64_t lo = 0, hi = ix->n_keys;
        while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ix->key[mid] < keys[t]) lo = mid + 1; else hi = mid; }
        counts[t] = (lo < ix->n_keys && ix->key[lo] == keys[t]) ? (uint32_t)(ix->off[lo + 1] - ix->off[lo]) : 0u;
    }
}

uint64_t lo_index_dump_minimizers(const lo_index_t *ix, lo_mm128_t *out, uint64_t cap)
{
    uint64_t n = ix->n_mz < cap ? ix->n_mz : cap;
    if (!ix->mz) return 0;      /* a restricted index (lo_ridx_finish) keeps no sketch-order copy */
    if (out) memcpy(out, ix->mz, n * sizeof(lo_mm128_t));
    return ix->n_mz;
}

static int cmp_u64(const void *pa, const void *pb);
static void set_threads(int threads);
/* ------------------------------------------------------------------------------------------ */
/* Index statistics of a target set too large to index here in one piece (H. sapiens-scale: 30 Gbases): the reads go  */
/* through mm_sketch chunk by chunk, only the minimizer hashes are kept, and n_minimizers, n_keys and mid_occ follow   */
/* with the arithmetic of calc_mid_occ above (the k-th smallest occurrence count over the distinct keys).              */
/* ------------------------------------------------------------------------------------------ */
struct lo_kstat { lo_opt_t opt; uint64_t *keys; uint64_t n, m; };

lo_kstat_t *lo_kstat_new(const lo_opt_t *opt)
{
    lo_kstat_t *s = (lo_kstat_t *)calloc(1, sizeof(*s));
    s->opt = *opt;
    return s;
}
void lo_kstat_free(lo_kstat_t *s) { if (s) { free(s->keys); free(s); } }

int lo_kstat_add(lo_kstat_t *s, const char *bases, const uint64_t *offs, uint32_t n, int threads)
{
    vec128_t *per = (vec128_t *)calloc(n ? n : 1, sizeof(vec128_t));
    uint64_t *start = (uint64_t *)malloc(((size_t)n + 1) * 8), tot = 0;
    int64_t r;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 16)
    for (r = 0; r < (int64_t)n; ++r)
        if (offs[r + 1] > offs[r])
            sketch_into(bases + offs[r], (int32_t)(offs[r + 1] - offs[r]), s->opt.w, s->opt.k, (uint32_t)r, s->opt.is_hpc, &per[r]);
    for (r = 0; r < (int64_t)n; ++r) { start[r] = tot; tot += (uint64_t)per[r].n; }
    if (s->n + tot > s->m) {
        s->m = (s->n + tot) * 5 / 4 + 1024;
        s->keys = (uint64_t *)realloc(s->keys, s->m * 8);
        if (!s->keys) return -1;
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (r = 0; r < (int64_t)n; ++r) {
        int64_t t;
        for (t = 0; t < per[r].n; ++t) s->keys[s->n + start[r] + (uint64_t)t] = per[r].a[t].x >> 8;
        free(per[r].a);
    }
    s->n += tot;
    free(per); free(start);
    return 0;
}

int lo_kstat_finish(lo_kstat_t *s, int threads, uint64_t *n_mz, uint64_t *n_keys, int32_t *mid_occ)
{
    enum { NB = 4096, HB = 1 << 20 };
    const int shift = 2 * s->opt.k > 12 ? 2 * s->opt.k - 12 : 0;
    uint64_t *bstart = (uint64_t *)calloc(NB + 2, 8), *fill, *srt, *hist, nk = 0, i;
    int64_t b, ii;
    int32_t thres;
    if (threads > 0) omp_set_num_threads(threads);
    srt = (uint64_t *)malloc((s->n ? s->n : 1) * 8);
    hist = (uint64_t *)calloc(HB + 1, 8);
    if (!srt || !hist) return -1;
#pragma omp parallel
    {
        uint64_t *loc = (uint64_t *)calloc(NB, 8);
        int64_t t;
#pragma omp for schedule(static) nowait
        for (t = 0; t < (int64_t)s->n; ++t) ++loc[s->keys[t] >> shift];
#pragma omp critical
        for (t = 0; t < NB; ++t) bstart[t + 1] += loc[t];
        free(loc);
    }
    for (i = 0; i < NB; ++i) bstart[i + 1] += bstart[i];
    fill = (uint64_t *)malloc((NB + 1) * 8);
    memcpy(fill, bstart, (NB + 1) * 8);
#pragma omp parallel for schedule(static)
    for (ii = 0; ii < (int64_t)s->n; ++ii) {
        uint64_t d;
#pragma omp atomic capture
        d = fill[s->keys[ii] >> shift]++;
        srt[d] = s->keys[ii];
    }
    free(fill);
#pragma omp parallel
    {
        uint64_t *lh = (uint64_t *)calloc(HB + 1, 8), lk = 0;
        int64_t t;
#pragma omp for schedule(dynamic, 8) nowait
        for (b = 0; b < NB; ++b) {
            uint64_t lo = bstart[b], hi = bstart[b + 1], p, run = 0;
            if (hi <= lo) continue;
            qsort(srt + lo, hi - lo, 8, cmp_u64);
            for (p = lo; p < hi; ++p) {
                ++run;
                if (p + 1 == hi || srt[p + 1] != srt[p]) { ++lk; ++lh[run < HB ? run : HB]; run = 0; }
            }
        }
#pragma omp critical
        { nk += lk; for (t = 0; t <= HB; ++t) hist[t] += lh[t]; }
        free(lh);
    }
    free(srt); free(bstart);
    /* calc_mid_occ: the k-th smallest count (0-based k = (1 - f) n) + 1, then the clamps of mm_mapopt_update */
    if (s->opt.mid_occ_frac <= 0.f || nk == 0) thres = INT32_MAX;
    else {
        const uint64_t kth = (uint64_t)((1. - (double)s->opt.mid_occ_frac) * (double)nk);
        uint64_t cum = 0, c;
        thres = -1;
        for (c = 0; c <= HB; ++c) { cum += hist[c]; if (cum > kth) { thres = (int32_t)c + 1; break; } }
        if (thres < 0 || thres > HB) { free(hist); return -2; }      /* the k-th count lies in the overflow bin */
    }
    free(hist);
    if (thres < s->opt.min_mid_occ) thres = s->opt.min_mid_occ;
    if (s->opt.max_mid_occ > s->opt.min_mid_occ && thres > s->opt.max_mid_occ) thres = s->opt.max_mid_occ;
    *n_mz = s->n; *n_keys = nk; *mid_occ = thres;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Restricted index: the index of a target set too large to hold here (H. sapiens-scale: 7.5 G minimizers), reduced   */
/* to what a SAMPLE of query reads can ask for.  mm_idx_get (index.c) is only ever called with the minimizers of the   */
/* query at hand (seed.c:mm_seed_collect_all), and its answer is the key's complete position list; an index that holds  */
/* the complete lists of exactly the keys occurring in the sample answers every one of those calls as the full index   */
/* would.  What the lists cannot tell is mid_occ (a statistic over ALL keys: mm_idx_cal_max_occ): it is handed in     */
/* (lo_kstat_* computes it for the same set).  The targets are fed chunk by chunk; every read keeps its rid, name and  */
/* length (skip_seed and the PAF fields read them).                                                                     */
/* ------------------------------------------------------------------------------------------ */
struct lo_ridx {
    lo_opt_t opt;
    uint64_t *tab, tab_mask;          /* open-addressing set of the sample's hashes (value + 1; 0 = empty) */
    uint64_t n_sample_keys;
    hy_t *hy; uint64_t n, m;          /* kept entries, any order */
    uint64_t n_mz_seen;               /* minimizers of all targets fed so far */
    uint32_t n_seq, m_seq;
    char **name; int32_t *len;
};

static inline uint64_t ridx_mix(uint64_t h) { h ^= h >> 31; h *= 0x9E3779B97F4A7C15ULL; h ^= h >> 29; return h; }

static int ridx_has(const lo_ridx_t *r, uint64_t h)
{
    uint64_t i = ridx_mix(h) & r->tab_mask;
    for (;;) {
        const uint64_t e = r->tab[i];
        if (e == 0) return 0;
        if (e == h + 1) return 1;
        i = (i + 1) & r->tab_mask;
    }
}

lo_ridx_t *lo_ridx_new(const lo_opt_t *opt, const char *qbases, const uint64_t *qoffs, uint32_t nq)
{
    lo_ridx_t *r = (lo_ridx_t *)calloc(1, sizeof(*r));
    vec128_t v = {0, 0, 0};
    uint64_t cap = 1024, i;
    uint32_t q;
    r->opt = *opt;
    for (q = 0; q < nq; ++q)
        if (qoffs[q + 1] > qoffs[q])
            sketch_into(qbases + qoffs[q], (int32_t)(qoffs[q + 1] - qoffs[q]), opt->w, opt->k, q, opt->is_hpc, &v);
    while (cap < 4 * (uint64_t)v.n + 16) cap <<= 1;
    r->tab = (uint64_t *)calloc(cap, 8);
    r->tab_mask = cap - 1;
    for (i = 0; i < (uint64_t)v.n; ++i) {
        const uint64_t h = v.a[i].x >> 8;
        uint64_t j = ridx_mix(h) & r->tab_mask;
        while (r->tab[j] != 0 && r->tab[j] != h + 1) j = (j + 1) & r->tab_mask;
        if (r->tab[j] == 0) { r->tab[j] = h + 1; ++r->n_sample_keys; }
    }
    free(v.a);
    return r;
}

void lo_ridx_free(lo_ridx_t *r)
{
    uint32_t i;
    if (!r) return;
    for (i = 0; i < r->n_seq; ++i) free(r->name[i]);
    free(r->name); free(r->len); free(r->hy); free(r->tab); free(r);
}

uint64_t lo_ridx_n_sample_keys(const lo_ridx_t *r) { return r->n_sample_keys; }
uint64_t lo_ridx_n_minimizers_seen(const lo_ridx_t *r) { return r->n_mz_seen; }
uint64_t lo_ridx_n_kept(const lo_ridx_t *r) { return r->n; }

/* the next n target reads (rid = reads fed before + index in this call) */
int lo_ridx_add(lo_ridx_t *r, const char *bases, const uint64_t *offs, uint32_t n, const char *const *names, int threads)
{
    const uint32_t rid0 = r->n_seq;
    int64_t i;
    int fail = 0;
    if ((uint64_t)rid0 + n > 0x7fffffffULL) return -3;
    if (rid0 + n > r->m_seq) {
        r->m_seq = (rid0 + n) * 2 + 1024;
        r->name = (char **)realloc(r->name, (size_t)r->m_seq * sizeof(char *));
        r->len = (int32_t *)realloc(r->len, (size_t)r->m_seq * sizeof(int32_t));
        if (!r->name || !r->len) return -1;
    }
    for (i = 0; i < (int64_t)n; ++i) {
        const char *nm = names && names[i] ? names[i] : "";
        r->name[rid0 + i] = (char *)malloc(strlen(nm) + 1);
        strcpy(r->name[rid0 + i], nm);
        r->len[rid0 + i] = (int32_t)(offs[i + 1] - offs[i]);
    }
    r->n_seq = rid0 + n;
    set_threads(threads);
#pragma omp parallel
    {
        vec128_t v = {0, 0, 0};
        hy_t *loc = 0; uint64_t ln = 0, lm = 0, seen = 0;
#pragma omp for schedule(dynamic, 16) nowait
        for (i = 0; i < (int64_t)n; ++i) {
            int64_t t;
            if (offs[i + 1] <= offs[i]) continue;
            v.n = 0;
            sketch_into(bases + offs[i], (int32_t)(offs[i + 1] - offs[i]), r->opt.w, r->opt.k, rid0 + (uint32_t)i, r->opt.is_hpc, &v);
            seen += (uint64_t)v.n;
            for (t = 0; t < v.n; ++t) {
                const uint64_t h = v.a[t].x >> 8;
                if (!ridx_has(r, h)) continue;
                if (ln == lm) { lm = lm ? lm * 2 : 4096; loc = (hy_t *)realloc(loc, lm * sizeof(hy_t)); if (!loc) { fail = 1; lm = ln = 0; break; } }
                loc[ln].h = h; loc[ln].y = v.a[t].y; ++ln;
            }
        }
#pragma omp critical
        {
            r->n_mz_seen += seen;
            if (ln) {
                if (r->n + ln > r->m) { r->m = (r->n + ln) * 3 / 2 + 4096; r->hy = (hy_t *)realloc(r->hy, r->m * sizeof(hy_t)); if (!r->hy) { fail = 1; r->m = r->n = 0; } }
                if (r->hy) { memcpy(r->hy + r->n, loc, ln * sizeof(hy_t)); r->n += ln; }
            }
        }
        free(loc); free(v.a);
    }
    return fail ? -1 : 0;
}

/* -> an index usable with lo_map / lo_twoset_counts / lo_anchors for the SAMPLE queries (and for nothing else);
   mid_occ: that of the whole target set (> 0).  The builder is consumed (and freed). */
lo_index_t *lo_ridx_finish(lo_ridx_t *r, lo_opt_t *opt, int32_t mid_occ, int threads)
{
    lo_index_t *ix;
    nameidx_t *ni;
    uint64_t i, j, n = r->n;
    hy_t *hy = r->hy;
    const uint32_t ns = r->n_seq;
    if (mid_occ <= 0) return 0;
    set_threads(threads);
    ix = (lo_index_t *)calloc(1, sizeof(*ix));
    ix->n_seq = ns; ix->k = r->opt.k; ix->w = r->opt.w; ix->is_hpc = r->opt.is_hpc;
    ix->name = r->name; ix->len = r->len; r->name = 0; r->len = 0; r->n_seq = 0;
    if (!ix->name) { ix->name = (char **)calloc(1, sizeof(char *)); ix->len = (int32_t *)calloc(1, sizeof(int32_t)); }
    ix->name_rank = (uint32_t *)calloc(ns ? ns : 1, sizeof(uint32_t));
    ni = (nameidx_t *)malloc((ns ? ns : 1) * sizeof(nameidx_t));
    for (i = 0; i < ns; ++i) { ni[i].s = ix->name[i]; ni[i].i = (uint32_t)i; }
    qsort(ni, ns, sizeof(nameidx_t), cmp_name);
    for (i = 0, j = 0; i < ns; ++i) {
        if (i > 0 && strcmp(ni[i].s, ni[i - 1].s) != 0) j = i;
        ix->name_rank[ni[i].i] = (uint32_t)j;
    }
    free(ni);
    {   /* (hash, y) is a total order on the entries (y is unique): bucket on the top 12 hash bits, sort the buckets in parallel */
        const int shift = 2 * r->opt.k > 12 ? 2 * r->opt.k - 12 : 0;
        uint64_t *bstart = (uint64_t *)calloc(4097, 8), *fill = (uint64_t *)malloc(4097 * 8);
        hy_t *srt = (hy_t *)malloc((n ? n : 1) * sizeof(hy_t));
        int64_t b;
        for (i = 0; i < n; ++i) ++bstart[(hy[i].h >> shift) + 1];
        for (i = 0; i < 4096; ++i) bstart[i + 1] += bstart[i];
        memcpy(fill, bstart, 4097 * 8);
        for (i = 0; i < n; ++i) srt[fill[hy[i].h >> shift]++] = hy[i];
        free(fill); free(hy); r->hy = 0; hy = srt;
#pragma omp parallel for schedule(dynamic, 8)
        for (b = 0; b < 4096; ++b)
            if (bstart[b + 1] > bstart[b]) qsort(hy + bstart[b], bstart[b + 1] - bstart[b], sizeof(hy_t), cmp_hy);
        free(bstart);
    }
    ix->n_mz = n; ix->mz = 0;
    for (i = 0, j = 0; i < n; ++i) j += (i == 0 || hy[i].h != hy[i - 1].h);
    ix->n_keys = j;
    ix->key = (uint64_t *)malloc((j ? j : 1) * 8);
    ix->off = (uint64_t *)malloc((j + 1) * 8);
    ix->pos = (uint64_t *)malloc((n ? n : 1) * 8);
    for (i = 0, j = 0; i < n; ++i) {
        if (i == 0 || hy[i].h != hy[i - 1].h) { ix->key[j] = hy[i].h; ix->off[j] = i; ++j; }
        ix->pos[i] = hy[i].y;
    }
    ix->off[j] = n;
    free(hy);
    opt->mid_occ = mid_occ;
    if (opt->bw_long < opt->bw) opt->bw_long = opt->bw;
    ix->mid_occ = mid_occ;
    lo_ridx_free(r);
    return ix;
}

int32_t lo_index_get(const lo_index_t *ix, uint64_t minier, const uint64_t **list)
{
    uint64_t lo = 0, hi = ix->n_keys;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (ix->key[mid] < minier) lo = mid + 1; else hi = mid;
    }
    if (lo == ix->n_keys || ix->key[lo] != minier) { if (list) *list = 0; return 0; }
    if (list) *list = ix->pos + ix->off[lo];
    if (ix->dropped && ix->dropped[lo]) return ix->mid_occ + 1;   /* (callers never walk a list of more than mid_occ entries) */
    return (int32_t)(ix->off[lo + 1] - ix->off[lo]);
}

/* This index holds a SHARD of a larger target set (the target-sharded multi-GPU form, lrge_hip_index_build_tsharded): keys whose
   occurrence count over the WHOLE set exceeds mid_occ answer as too frequent here too, whatever their count in the shard --
   what mm_idx_get would say in the one index.  Returns how many of the keys the shard holds. */
uint64_t lo_index_drop_keys(lo_index_t *ix, const uint64_t *keys, uint64_t n)
{
    uint64_t i, found = 0;
    if (!ix->dropped) ix->dropped = (unsigned char *)calloc(ix->n_keys ? ix->n_keys : 1, 1);
    for (i = 0; i < n; ++i) {
        uint64_t lo = 0, hi = ix->n_keys;
        while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ix->key[mid] < keys[i]) lo = mid + 1; else hi = mid; }
        if (lo < ix->n_keys && ix->key[lo] == keys[i]) { ix->dropped[lo] = 1; ++found; }
    }
    return found;
}

/* ------------------------------------------------------------------------------------------ */
/* seeding: mm2:seed.c:mm_seed_mz_flt, mm_seed_collect_all, mm_collect_matches (occ_dist = 0); */
/* mm2:map.c:skip_seed, collect_seed_hits                                                      */
/* ------------------------------------------------------------------------------------------ */
#define SEED_TANDEM (1ULL << 42)
#define SEED_SELF   (1ULL << 43)

typedef struct {
    uint32_t n, q_pos, q_span : 31, flt : 1;
    uint32_t is_tandem;
    const uint64_t *cr;
} seed_t;

static void seed_mz_flt(vec128_t *mv, int32_t q_occ_max, float q_occ_frac, int sort_mode)
{
    lo_mm128_t *a;
    int64_t i, j, st, n = mv->n;
    if (n <= q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return;
    a = (lo_mm128_t *)malloc((size_t)n * sizeof(lo_mm128_t));
    for (i = 0; i < n; ++i) { a[i].x = mv->a[i].x; a[i].y = (uint64_t)i; }
    lo_sort128x(a, n, sort_mode);
    for (st = 0, i = 1; i <= n; ++i) {
        if (i == n || a[i].x != a[st].x) {
            int32_t cnt = (int32_t)(i - st);
            if (cnt > q_occ_max && (float)cnt > (float)(uint64_t)n * q_occ_frac)
                for (j = st; j < i; ++j) mv->a[a[j].y].x = 0;
            st = i;
        }
    }
    free(a);
    for (i = j = 0; i < n; ++i)
        if (mv->a[i].x != 0) mv->a[j++] = mv->a[i];
    mv->n = j;
}

typedef struct {
    lo_mm128_t *a; int64_t n_a;          /* sorted anchors */
    uint64_t *mini_pos; int32_t n_mini_pos;
    int32_t rep_len;
} seeds_out_t;

static void collect_seed_hits(const lo_index_t *ix, const lo_opt_t *opt, const char *seq,
                              int32_t qlen, const char *qname, seeds_out_t *so)
{
    vec128_t mv = {0, 0, 0};
    seed_t *m;
    int64_t i, n_m0 = 0, n_m = 0, n_a = 0;
    int32_t rep_st = 0, rep_en = 0, rep_len = 0;
    lo_mm128_t *a;

    memset(so, 0, sizeof(*so));
    sketch_into(seq, qlen, ix->w, ix->k, 0, ix->is_hpc, &mv); /* collect_minimizers: rid = seg 0 */
    if (opt->q_occ_frac > 0.0f) seed_mz_flt(&mv, opt->mid_occ, opt->q_occ_frac, opt->sort_mode);

    m = (seed_t *)malloc((size_t)(mv.n ? mv.n : 1) * sizeof(seed_t));
    so->mini_pos = (uint64_t *)malloc((size_t)(mv.n ? mv.n : 1) * 8);
    for (i = 0; i < mv.n; ++i) { /* mm_seed_collect_all */
        const uint64_t *cr;
        lo_mm128_t *p = &mv.a[i];
        int32_t t = lo_index_get(ix, p->x >> 8, &cr);
        seed_t *q;
        if (t == 0) continue;
        q = &m[n_m0++];
        q->q_pos = (uint32_t)p->y; q->q_span = (uint32_t)(p->x & 0xff); q->cr = cr; q->n = (uint32_t)t;
        q->is_tandem = 0; q->flt = 0;
        if (i > 0 && p->x >> 8 == mv.a[i - 1].x >> 8) q->is_tandem = 1;
        if (i < mv.n - 1 && p->x >> 8 == mv.a[i + 1].x >> 8) q->is_tandem = 1;
    }
    for (i = 0; i < n_m0; ++i) /* occ_dist == 0 branch of mm_collect_matches */
        if ((int64_t)m[i].n > (int64_t)opt->mid_occ) m[i].flt = 1;
    for (i = 0; i < n_m0; ++i) {
        seed_t *q = &m[i];
        if (q->flt) {
            int32_t en = (int32_t)(q->q_pos >> 1) + 1, st = en - (int32_t)q->q_span;
            if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
            else rep_en = en;
        } else {
            n_a += q->n;
            so->mini_pos[so->n_mini_pos++] = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
            m[n_m++] = *q;
        }
    }
    rep_len += rep_en - rep_st;
    so->rep_len = rep_len;

    a = (lo_mm128_t *)malloc((size_t)(n_a ? n_a : 1) * sizeof(lo_mm128_t));
    n_a = 0;
    for (i = 0; i < n_m; ++i) {
        seed_t *q = &m[i];
        uint32_t kk;
        for (kk = 0; kk < q->n; ++kk) {
            uint64_t r = q->cr[kk];
            int32_t rpos = (int32_t)((uint32_t)r >> 1), is_self = 0;
            lo_mm128_t *p;
            /* skip_seed */
            if (qname && (opt->flag & (LO_F_NO_DIAG | LO_F_NO_DUAL))) {
                uint32_t rid = (uint32_t)(r >> 32);
                int cmp = strcmp(qname, ix->name[rid]);
                if ((opt->flag & LO_F_NO_DIAG) && cmp == 0 && ix->len[rid] == qlen) {
                    if ((uint32_t)r >> 1 == (q->q_pos >> 1)) continue; /* exact diagonal */
                    if ((r & 1) == (q->q_pos & 1)) is_self = 1;
                }
                if ((opt->flag & LO_F_NO_DUAL) && cmp > 0) continue;
            }
            p = &a[n_a++];
            if ((r & 1) == (q->q_pos & 1)) { /* same strand */
                p->x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
            } else { /* opposite strand: query coordinate flipped */
                p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
                p->y = (uint64_t)q->q_span << 32 |
                       (uint32_t)(qlen - ((int32_t)(q->q_pos >> 1) + 1 - (int32_t)q->q_span) - 1);
            }
            if (q->is_tandem) p->y |= SEED_TANDEM;
            if (is_self) p->y |= SEED_SELF;
        }
    }
    free(m); free(mv.a);
    lo_sort128x(a, n_a, opt->sort_mode); /* radix_sort_128x: ascending by x only */
    so->a = a; so->n_a = n_a;
}

int64_t lo_anchors(const lo_index_t *ix, const lo_opt_t *opt, const char *seq, int32_t qlen,
                   const char *qname, lo_mm128_t *out, int64_t cap)
{
    seeds_out_t so;
    int64_t n;
    collect_seed_hits(ix, opt, seq, qlen, qname, &so);
    n = so.n_a;
    if (out) memcpy(out, so.a, (size_t)(n < cap ? n : cap) * sizeof(lo_mm128_t));
    free(so.a); free(so.mini_pos);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* chaining: mm2:mmpriv.h:mg_log2; mm2:lchain.c:comput_sc, mg_lchain_dp, mg_chain_bk_end,      */
/* mg_chain_backtrack, compact_a                                                               */
/* ------------------------------------------------------------------------------------------ */
static inline float mg_log2f(float x) /* only meaningful for x >= 2 */
{
    union { float f; uint32_t i; } z;
    float log_2;
    z.f = x;
    log_2 = (float)(int32_t)(((z.i >> 23) & 255) - 128);
    z.i &= ~(255U << 23);
    z.i += 127U << 23;
    log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
    return log_2;
}

static inline int32_t comput_sc(const lo_mm128_t *ai, const lo_mm128_t *aj, int32_t max_dist_x,
                                int32_t max_dist_y, int32_t bw, float chn_pen_gap, float chn_pen_skip)
{
    int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
    if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
    dr = (int32_t)(ai->x - aj->x);
    if (dr == 0 || dq > max_dist_y) return INT32_MIN; /* single segment: sidi == sidj */
    dd = dr > dq ? dr - dq : dq - dr;
    if (dd > bw) return INT32_MIN;
    dg = dr < dq ? dr : dq;
    q_span = (int32_t)(aj->y >> 32 & 0xff);
    sc = q_span < dg ? q_span : dg;
    if (dd || dg > q_span) {
        float lin_pen, log_pen;
        lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
        log_pen = dd >= 1 ? mg_log2f((float)(dd + 1)) : 0.0f;
        sc -= (int)(lin_pen + .5f * log_pen);
    }
    return sc;
}

static int64_t chain_bk_end(int32_t max_drop, const lo_mm128_t *z, const int32_t *f,
                            const int64_t *p, int32_t *t, int64_t k)
{
    int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
    int32_t max_s = 0;
    if (i < 0 || t[i] != 0) return i;
    do {
        int32_t s;
        t[i] = 2;
        end_i = i = p[i];
        s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
        if (s > max_s) { max_s = s; max_i = i; }
        else if (max_s - s > max_drop) break;
    } while (i >= 0 && t[i] == 0);
    for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
    return max_i;
}

/* returns chains in a[] (compacted, each chain ascending, chains ordered by first-anchor x),
   u[i] = score<<32 | n_anchors.  The two-pass count/fill of mg_chain_backtrack is done once. */
static lo_mm128_t *lchain_dp(const lo_opt_t *opt, int32_t kmer, int64_t n, lo_mm128_t *a,
                             int32_t *n_u_, uint64_t **u_)
{
    int32_t max_dist_x = opt->max_gap_ref > 0 ? opt->max_gap_ref : opt->max_gap; /* max_chain_gap_ref */
    int32_t max_dist_y = opt->max_gap;                                            /* max_chain_gap_qry */
    const int32_t bw = opt->bw, max_skip = opt->max_chain_skip, max_iter = opt->max_chain_iter;
    const int32_t min_cnt = opt->min_cnt, min_sc = opt->min_chain_score, max_drop = bw;
    /* chn_pen_* = scale * 0.01 * k evaluated in double, then narrowed (mm2:map.c:mm_map_frag) */
    const float chn_pen_gap = (float)((double)opt->chain_gap_scale * 0.01 * (double)kmer);
    const float chn_pen_skip = (float)((double)opt->chain_skip_scale * 0.01 * (double)kmer);
    int32_t *f, *t, *v, n_u = 0;
    int64_t *p, i, j, k, max_ii, st = 0, n_z, n_v;
    uint64_t *u;
    lo_mm128_t *z, *b, *wv;

    *n_u_ = 0; *u_ = 0;
    if (n == 0 || a == 0) { free(a); return 0; }
    if (max_dist_x < bw) max_dist_x = bw;
    if (max_dist_y < bw) max_dist_y = bw;
    p = (int64_t *)malloc((size_t)n * 8);
    f = (int32_t *)malloc((size_t)n * 4);
    v = (int32_t *)malloc((size_t)n * 4);
    t = (int32_t *)calloc((size_t)n, 4);

    for (i = 0, max_ii = -1; i < n; ++i) {
        int64_t max_j = -1, end_j;
        int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
        while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist_x)) ++st;
        if (i - st > max_iter) st = i - max_iter;
        for (j = i - 1; j >= st; --j) {
            int32_t sc = comput_sc(&a[i], &a[j], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip);
            if (sc == INT32_MIN) continue;
            sc += f[j];
            if (sc > max_f) {
                max_f = sc; max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == (int32_t)i) {
                if (++n_skip > max_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = (int32_t)i;
        }
        end_j = j;
        if (max_ii < 0 || a[i].x - a[max_ii].x > (uint64_t)(int64_t)max_dist_x) {
            int32_t mx = INT32_MIN;
            max_ii = -1;
            for (j = i - 1; j >= st; --j)
                if (mx < f[j]) { mx = f[j]; max_ii = j; }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            int32_t tmp = comput_sc(&a[i], &a[max_ii], max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip);
            if (tmp != INT32_MIN && max_f < tmp + f[max_ii]) { max_f = tmp + f[max_ii]; max_j = max_ii; }
        }
        f[i] = max_f; p[i] = max_j;
        if (max_ii < 0 || (a[i].x - a[max_ii].x <= (uint64_t)(int64_t)max_dist_x && f[max_ii] < f[i]))
            max_ii = i;
    }

    /* mg_chain_backtrack */
    for (i = 0, n_z = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
    if (n_z == 0) { free(p); free(f); free(t); free(v); free(a); return 0; }
    z = (lo_mm128_t *)malloc((size_t)n_z * sizeof(lo_mm128_t));
    for (i = 0, k = 0; i < n; ++i)
        if (f[i] >= min_sc) { z[k].x = (uint64_t)f[i]; z[k++].y = (uint64_t)i; }
    lo_sort128x(z, n_z, opt->sort_mode);
    memset(t, 0, (size_t)n * 4);
    u = (uint64_t *)malloc((size_t)n_z * 8);
    for (k = n_z - 1, n_v = 0, n_u = 0; k >= 0; --k) {
        if (t[z[k].y] == 0) {
            int64_t n_v0 = n_v, end_i;
            int32_t sc;
            end_i = chain_bk_end(max_drop, z, f, p, t, k);
            for (i = (int64_t)z[k].y; i != end_i; i = p[i]) { v[n_v++] = (int32_t)i; t[i] = 1; }
            sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
            if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt)
                u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
            else n_v = n_v0;
        }
    }
    free(z); free(p); free(f); free(t);
    if (n_u == 0) { free(a); free(v); free(u); return 0; }

    /* compact_a */
    b = (lo_mm128_t *)malloc((size_t)n_v * sizeof(lo_mm128_t));
    for (i = 0, k = 0; i < n_u; ++i) {
        int64_t k0 = k; int32_t ni = (int32_t)u[i];
        for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
    }
    free(v);
    wv = (lo_mm128_t *)malloc((size_t)n_u * sizeof(lo_mm128_t));
    for (i = k = 0; i < n_u; ++i) {
        wv[i].x = b[k].x; wv[i].y = (uint64_t)k << 32 | (uint64_t)i;
        k += (int32_t)u[i];
    }
    lo_sort128x(wv, n_u, opt->sort_mode);
    {
        uint64_t *u2 = (uint64_t *)malloc((size_t)n_u * 8);
        for (i = k = 0; i < n_u; ++i) {
            int32_t jj = (int32_t)wv[i].y, nn = (int32_t)u[jj];
            u2[i] = u[jj];
            memcpy(&a[k], &b[wv[i].y >> 32], (size_t)nn * sizeof(lo_mm128_t));
            k += nn;
        }
        memcpy(u, u2, (size_t)n_u * 8);
        free(u2);
    }
    free(b); free(wv);
    *n_u_ = n_u; *u_ = u;
    return a; /* first n_v entries hold the chains */
}

/* ------------------------------------------------------------------------------------------ */
/* regions: mm2:hit.c:mm_gen_regs, mm_reg_set_coor, mm_cal_fuzzy_len; mm2:esterr.c:mm_est_err  */
/* field use: aligner.rs:244-291                                                              */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t wang64(uint64_t key)
{
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}
static inline uint32_t wang32(uint32_t key)
{
    key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
    key ^= (key >> 6);   key += ~(key << 11); key ^= (key >> 16);
    return key;
}
static inline uint32_t x31_hash(const char *s)
{
    uint32_t h = (uint32_t)*s;
    if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
    return h;
}

static inline int32_t fwd_qpos(int32_t qlen, const lo_mm128_t *a)
{
    int32_t x = (int32_t)a->y, q_span = (int32_t)(a->y >> 32 & 0xff);
    if (a->x >> 63) x = qlen - 1 - (x + 1 - q_span);
    return x;
}

static int32_t gen_regs(const lo_index_t *ix, const lo_opt_t *opt, int32_t qlen, const char *qname,
                        int32_t n_u, const uint64_t *u, const lo_mm128_t *a, const seeds_out_t *so,
                        lo_reg_t *out, int32_t cap)
{
    lo_mm128_t *z;
    uint32_t hash;
    int32_t i, k;
    float avg_k = 0.f;
    if (n_u == 0) return 0;
    hash = qname ? x31_hash(qname) : 0;
    hash ^= wang32((uint32_t)qlen) + wang32((uint32_t)opt->seed);
    hash = wang32(hash);
    z = (lo_mm128_t *)malloc((size_t)n_u * sizeof(lo_mm128_t));
    for (i = k = 0; i < n_u; ++i) {
        uint32_t h = (uint32_t)wang64((wang64(a[k].x) + wang64(a[k].y)) ^ hash);
        z[i].x = u[i] ^ h;
        z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
        k += (int32_t)u[i];
    }
    lo_sort128x(z, n_u, opt->sort_mode);
    for (i = 0; i < n_u >> 1; ++i) { lo_mm128_t tmp = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = tmp; }

    if (so->n_mini_pos > 0) {
        uint64_t sum_k = 0;
        for (i = 0; i < so->n_mini_pos; ++i) sum_k += so->mini_pos[i] >> 32 & 0xff;
        avg_k = (float)sum_k / (float)so->n_mini_pos;
    }
    for (i = 0; i < n_u && i < cap; ++i) {
        lo_reg_t *r = &out[i];
        int32_t as = (int32_t)(z[i].y >> 32), cnt = (int32_t)z[i].y, q_span, j;
        const lo_mm128_t *c = a + as;
        memset(r, 0, sizeof(*r));
        r->score = (int32_t)(z[i].x >> 32);
        r->cnt = cnt;
        /* mm_reg_set_coor */
        q_span = (int32_t)(c[0].y >> 32 & 0xff);
        r->rev = (int32_t)(c[0].x >> 63);
        r->rid = (int32_t)(c[0].x << 1 >> 33);
        r->rs = (int32_t)c[0].x + 1 > q_span ? (int32_t)c[0].x + 1 - q_span : 0;
        r->re = (int32_t)c[cnt - 1].x + 1;
        if (!r->rev) {
            r->qs = (int32_t)c[0].y + 1 - q_span;
            r->qe = (int32_t)c[cnt - 1].y + 1;
        } else {
            r->qs = qlen - ((int32_t)c[cnt - 1].y + 1);
            r->qe = qlen - ((int32_t)c[0].y + 1 - q_span);
        }
        /* mm_cal_fuzzy_len */
        r->mlen = r->blen = q_span;
        for (j = 1; j < cnt; ++j) {
            int32_t span = (int32_t)(c[j].y >> 32 & 0xff);
            int32_t tl = (int32_t)c[j].x - (int32_t)c[j - 1].x;
            int32_t ql = (int32_t)c[j].y - (int32_t)c[j - 1].y;
            r->blen += tl > ql ? tl : ql;
            r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
        }
        /* mm_est_err */
        r->dv = -1.0f;
        r->rep_len = so->rep_len;
        if (so->n_mini_pos > 0 && cnt > 0) {
            int32_t x = fwd_qpos(qlen, r->rev ? &c[cnt - 1] : &c[0]);
            int32_t L = 0, R = so->n_mini_pos - 1, st = -1, en, kk, n_match, n_tot, l_ref;
            while (L <= R) {
                int32_t mid = (int32_t)(((uint64_t)L + (uint64_t)R) >> 1);
                int32_t y = (int32_t)so->mini_pos[mid];
                if (y < x) L = mid + 1; else if (y > x) R = mid - 1; else { st = mid; break; }
            }
            if (st >= 0) {
                en = st; l_ref = ix->len[r->rid];
                for (kk = 1, j = st + 1, n_match = 1; j < so->n_mini_pos && kk < cnt; ++j) {
                    int32_t xx = fwd_qpos(qlen, r->rev ? &c[cnt - 1 - kk] : &c[kk]);
                    if (xx == (int32_t)so->mini_pos[j]) { ++kk; en = j; ++n_match; }
                }
                n_tot = en - st + 1;
                if ((float)r->qs > avg_k && (float)r->rs > avg_k) ++n_tot;
                if ((float)(qlen - r->qs) > avg_k && (float)(l_ref - r->re) > avg_k) ++n_tot;
                r->dv = n_match >= n_tot ? 0.0f
                        : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
            }
        }
    }
    free(z);
    return n_u;
}

/* mm2:map.c:mm_map_frag (n_segs = 1, NO_LJOIN, max_occ = 0, ALL_CHAINS => chain_post no-op,  */
/* no CIGAR => align_regs no-op); called through aligner.rs:231-241                           */
int32_t lo_map(const lo_index_t *ix, const lo_opt_t *opt, const char *seq, int32_t qlen,
               const char *qname, lo_reg_t *out, int32_t cap)
{
    seeds_out_t so;
    int32_t n_u = 0, n;
    uint64_t *u = 0;
    lo_mm128_t *a;
    if (qlen <= 0) return 0;
    collect_seed_hits(ix, opt, seq, qlen, qname, &so);
    a = lchain_dp(opt, ix->k, so.n_a, so.a, &n_u, &u); /* consumes so.a */
    n = gen_regs(ix, opt, qlen, qname, n_u, u, a, &so, out, cap);
    free(a); free(u); free(so.mini_pos);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* liblrge: mapping.rs:59-77 (is_internal), twoset.rs:493-517 (inverse inline predicate)       */
/* ------------------------------------------------------------------------------------------ */
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

static void overhang_maplen(int32_t qlen, int32_t qs, int32_t qe, int rev, int32_t tlen, int32_t ts,
                            int32_t te, int32_t *overhang, int32_t *maplen)
{
    if (!rev) *overhang = imin(qs, ts) + imin(qlen - qe, tlen - te);
    else *overhang = imin(qs, tlen - te) + imin(qlen - qe, ts);
    *maplen = imax(qe - qs, te - ts);
}

int lo_is_internal(int32_t qlen, int32_t qs, int32_t qe, int rev, int32_t tlen, int32_t ts,
                   int32_t te, float max_overhang_ratio)
{
    int32_t overhang, maplen;
    float ratio;
    overhang_maplen(qlen, qs, qe, rev, tlen, ts, te, &overhang, &maplen);
    ratio = (float)overhang / (float)maplen;
    return ratio < max_overhang_ratio;
}

int lo_inverse_skip(int32_t qlen, int32_t qs, int32_t qe, int rev, int32_t tlen, int32_t ts,
                    int32_t te, float max_overhang_ratio)
{
    int32_t overhang, maplen, lim;
    float prod;
    overhang_maplen(qlen, qs, qe, rev, tlen, ts, te, &overhang, &maplen);
    prod = (float)maplen * max_overhang_ratio;
    /* Rust `as i32`: saturating, NaN -> 0 */
    if (prod != prod) lim = 0;
    else if (prod >= 2147483648.0f) lim = INT32_MAX;
    else if (prod <= -2147483648.0f) lim = INT32_MIN;
    else lim = (int32_t)prod;
    return overhang > lim;
}

/* ------------------------------------------------------------------------------------------ */
/* counting shells: twoset.rs:266-334 (forward), twoset.rs:457-524 (inverse), ava.rs:243-306   */
/* ------------------------------------------------------------------------------------------ */
#define MAX_REGS_PER_QUERY 65536

/* Every query allocates a dozen arrays sized by its anchors (minimap2 takes them from kalloc, a per-thread arena: mm_tbuf_t,
   thread_buf.rs:7-42).  With glibc's defaults each one above 128 KB is its own mmap / munmap, and 256 mapping threads then queue
   for the process's address-space lock instead of mapping -- an artefact of this port, not of the algorithm.  Served from
   the threads' malloc arenas, never trimmed, the allocations behave like kalloc's.  LO_NO_MALLOPT=1 keeps glibc's defaults
   (the A/B of tools/cpu_port_scaling.py; on this pool's boxes, which grant 16 CPUs, the two measure alike). */
static void tune_malloc(void)
{
    static int done = 0;
    if (done) return;
    done = 1;
    if (getenv("LO_NO_MALLOPT")) return;
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, INT_MAX);
    mallopt(M_TOP_PAD, 64 << 20);
}

/* the thread count of every parallel region that is not given one (lo_index_build, lo_kstat_*, lo_ridx_*; threads <= 0 in the counting
   shells): oracle.py sets it to what the host GRANTS (cgroup CPU bandwidth), which on the GPU boxes of this pool is 16 CPUs, not the
   256 hardware threads omp_get_max_threads() reports */
void lo_set_default_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static void set_threads(int threads)
{
    tune_malloc();
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}

int lo_twoset_counts(const lo_index_t *ix, const lo_opt_t *opt, const char *qbases,
                     const uint64_t *qoffs, uint32_t nq, const char *const *qnames,
                     int remove_internal, float max_overhang_ratio, int threads,
                     uint32_t *counts, uint32_t *has_mapping)
{
    int64_t q;
    int err = 0;
    set_threads(threads);
#pragma omp parallel
    {
        lo_reg_t *regs = (lo_reg_t *)malloc(MAX_REGS_PER_QUERY * sizeof(lo_reg_t));
        uint32_t *stamp = (uint32_t *)calloc(ix->n_seq ? ix->n_seq : 1, 4); /* keyed by name rank */
#pragma omp for schedule(dynamic, 4)
        for (q = 0; q < (int64_t)nq; ++q) {
            int32_t qlen = (int32_t)(qoffs[q + 1] - qoffs[q]), n, i;
            uint32_t c = 0;
            if (qlen <= 0) { /* aligner.rs:214-216 "Sequence is empty" -> MapError aborts the run */
#pragma omp atomic write
                err = -6;
                counts[q] = 0; if (has_mapping) has_mapping[q] = 0;
                continue;
            }
            n = lo_map(ix, opt, qbases + qoffs[q], qlen, qnames ? qnames[q] : 0, regs, MAX_REGS_PER_QUERY);
            if (n > MAX_REGS_PER_QUERY) n = MAX_REGS_PER_QUERY;
            for (i = 0; i < n; ++i) {
                const lo_reg_t *r = &regs[i];
                uint32_t key = ix->name_rank[r->rid];
                if (remove_internal &&
                    lo_is_internal(qlen, r->qs, r->qe, r->rev, ix->len[r->rid], r->rs, r->re, max_overhang_ratio))
                    continue;
                if (stamp[key] != (uint32_t)q + 1) { stamp[key] = (uint32_t)q + 1; ++c; } /* HashSet of target_name */
            }
            counts[q] = c;
            if (has_mapping) has_mapping[q] = n > 0;
        }
        free(regs); free(stamp);
    }
    return err;
}

int lo_inverse_counts(const lo_index_t *ix, const lo_opt_t *opt, const char *tbases,
                      const uint64_t *toffs, uint32_t nt, const char *const *tnames,
                      int remove_internal, float max_overhang_ratio, int threads, uint32_t *counts)
{
    int64_t q;
    uint32_t i;
    int err = 0;
    /* duplicate identifiers among the indexed reads are a hard error (twoset.rs:439-449) */
    for (i = 0; i < ix->n_seq; ++i) counts[i] = 0;
    {
        uint32_t *seen = (uint32_t *)calloc(ix->n_seq ? ix->n_seq : 1, 4);
        for (i = 0; i < ix->n_seq; ++i) { if (seen[ix->name_rank[i]]++) err = -7; }
        free(seen);
        if (err) return err;
    }
    set_threads(threads);
#pragma omp parallel
    {
        lo_reg_t *regs = (lo_reg_t *)malloc(MAX_REGS_PER_QUERY * sizeof(lo_reg_t));
        uint32_t *stamp = (uint32_t *)calloc(ix->n_seq ? ix->n_seq : 1, 4);
#pragma omp for schedule(dynamic, 4)
        for (q = 0; q < (int64_t)nt; ++q) {
            int32_t qlen = (int32_t)(toffs[q + 1] - toffs[q]), n, j;
            if (qlen <= 0) {
#pragma omp atomic write
                err = -6;
                continue;
            }
            n = lo_map(ix, opt, tbases + toffs[q], qlen, tnames ? tnames[q] : 0, regs, MAX_REGS_PER_QUERY);
            if (n > MAX_REGS_PER_QUERY) n = MAX_REGS_PER_QUERY;
            for (j = 0; j < n; ++j) {
                const lo_reg_t *r = &regs[j];
                uint32_t key = (uint32_t)r->rid; /* names unique here, so rid identifies the name */
                if (stamp[key] == (uint32_t)q + 1) continue;
                if (remove_internal &&
                    lo_inverse_skip(qlen, r->qs, r->qe, r->rev, ix->len[r->rid], r->rs, r->re, max_overhang_ratio))
                    continue;
#pragma omp atomic
                counts[key] += 1;
                stamp[key] = (uint32_t)q + 1;
            }
        }
        free(regs); free(stamp);
    }
    return err;
}

static int cmp_u64(const void *pa, const void *pb)
{
    uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
    return a < b ? -1 : a > b;
}

int lo_ava_counts(const lo_index_t *ix, const lo_opt_t *opt, const char *bases,
                  const uint64_t *offs, uint32_t n, const char *const *names, int remove_internal,
                  float max_overhang_ratio, int threads, uint32_t *counts)
{
    int64_t q;
    uint32_t i;
    int err = 0, nth = 1, tid_total;
    uint64_t **pairs, *np, *cap, total = 0, *all, k;
    /* duplicate identifiers -> DuplicateReadIdentifier (ava.rs:195-199) */
    {
        uint32_t *seen = (uint32_t *)calloc(ix->n_seq ? ix->n_seq : 1, 4);
        for (i = 0; i < ix->n_seq; ++i) { if (seen[ix->name_rank[i]]++) err = -7; }
        free(seen);
        if (err) return err;
    }
    for (i = 0; i < n; ++i) counts[i] = 0;
    set_threads(threads);
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    tid_total = nth;
    pairs = (uint64_t **)calloc((size_t)nth, sizeof(uint64_t *));
    np = (uint64_t *)calloc((size_t)nth, 8);
    cap = (uint64_t *)calloc((size_t)nth, 8);
#pragma omp parallel
    {
        int tid = 0;
        lo_reg_t *regs = (lo_reg_t *)malloc(MAX_REGS_PER_QUERY * sizeof(lo_reg_t));
        uint32_t *stamp = (uint32_t *)calloc(ix->n_seq ? ix->n_seq : 1, 4);
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
#pragma omp for schedule(dynamic, 4)
        for (q = 0; q < (int64_t)n; ++q) {
            int32_t qlen = (int32_t)(offs[q + 1] - offs[q]), nr, j;
            if (qlen <= 0) {
#pragma omp atomic write
                err = -6;
                continue;
            }
            nr = lo_map(ix, opt, bases + offs[q], qlen, names[q], regs, MAX_REGS_PER_QUERY);
            if (nr > MAX_REGS_PER_QUERY) nr = MAX_REGS_PER_QUERY;
            for (j = 0; j < nr; ++j) {
                const lo_reg_t *r = &regs[j];
                uint32_t t = (uint32_t)r->rid;
                uint64_t a, b;
                if (strcmp(names[q], ix->name[t]) == 0) continue; /* self (ava.rs:277-281) */
                if (remove_internal &&
                    lo_is_internal(qlen, r->qs, r->qe, r->rev, ix->len[t], r->rs, r->re, max_overhang_ratio))
                    continue;
                if (stamp[t] == (uint32_t)q + 1) continue;
                stamp[t] = (uint32_t)q + 1;
                /* the query set IS the indexed set in AVA: read q has index id q */
                a = (uint64_t)q < t ? (uint64_t)q : t; b = (uint64_t)q < t ? t : (uint64_t)q;
                if (np[tid] == cap[tid]) {
                    cap[tid] = cap[tid] ? cap[tid] * 2 : 1024;
                    pairs[tid] = (uint64_t *)realloc(pairs[tid], cap[tid] * 8);
                }
                pairs[tid][np[tid]++] = a << 32 | b;
            }
        }
        free(regs); free(stamp);
    }
    for (i = 0; i < (uint32_t)tid_total; ++i) total += np[i];
    all = (uint64_t *)malloc((total ? total : 1) * 8);
    for (i = 0, k = 0; i < (uint32_t)tid_total; ++i) {
        if (np[i]) memcpy(all + k, pairs[i], np[i] * 8);
        k += np[i]; free(pairs[i]);
    }
    qsort(all, total, 8, cmp_u64);
    for (k = 0; k < total; ++k) { /* seen_pairs (ava.rs:289-298): each unordered pair once */
        if (k > 0 && all[k] == all[k - 1]) continue;
        counts[all[k] >> 32] += 1;
        counts[(uint32_t)all[k]] += 1;
    }
    free(all); free(pairs); free(np); free(cap);
    return err;
}

/* ------------------------------------------------------------------------------------------ */
/* estimate.rs:142-157 (per_read_estimate), :80-132 (median, calculate_quantile)               */
/* ------------------------------------------------------------------------------------------ */
float lo_per_read_estimate(uint64_t read_len, float avg_target_len, uint64_t n_target_reads,
                           uint64_t n_ovlaps, uint32_t ovlap_thresh)
{
    volatile float ratio, t; /* volatile: forbid contraction / reassociation */
    float rl = (float)read_len;
    if (n_ovlaps == 0) return INFINITY;
    ratio = (float)n_target_reads / (float)n_ovlaps;
    t = rl + avg_target_len;
    t = t - 2.0f * (float)ovlap_thresh;
    t = t + 1.0f;
    t = ratio * t;
    return rl + t;
}

static int cmp_f32(const void *pa, const void *pb)
{
    float a = *(const float *)pa, b = *(const float *)pb;
    return a < b ? -1 : a > b;
}

static int quantile(const float *d, uint64_t n, float q, float *out)
{
    volatile float pos, frac, lo, hi;
    uint64_t idx;
    if (n == 0) return 0;
    pos = q * (float)(n - 1);
    idx = (uint64_t)floorf(pos);
    frac = pos - (float)idx;
    if (idx + 1 < n) {
        lo = d[idx] * (1.0f - frac);
        hi = d[idx + 1] * frac;
        *out = lo + hi;
    } else *out = d[idx];
    return 1;
}

int lo_median(const float *vals, uint64_t n, int finite_only, int has_lower, float lower_q,
              int has_upper, float upper_q, float out[3], int ok[3])
{
    float *v = (float *)malloc((n ? n : 1) * sizeof(float));
    uint64_t i, m = 0;
    ok[0] = ok[1] = ok[2] = 0;
    out[0] = out[1] = out[2] = 0.f;
    if (!has_lower && has_upper) { free(v); return -1; } /* reference indexes out of bounds here */
    for (i = 0; i < n; ++i)
        if (!finite_only || isfinite(vals[i])) v[m++] = vals[i];
    if (m == 0) { free(v); return 0; }
    qsort(v, m, sizeof(float), cmp_f32);
    ok[1] = quantile(v, m, 0.5f, &out[1]);
    if (has_lower) ok[0] = quantile(v, m, lower_q, &out[0]);
    if (has_upper) ok[2] = quantile(v, m, upper_q, &out[2]);
    free(v);
    return 0;
}
