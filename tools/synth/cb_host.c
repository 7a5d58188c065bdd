/*
 * cb_host.c -- host twin of the counter-based read generator (cb_core.h).  TEST / BENCH INFRASTRUCTURE.
 * Any subset of the reads, in any order, on any number of threads: a read depends on (params, its index) only.
 */
#include <stdint.h>
#include <stddef.h>
#include "cb_core.h"

static const char CB_ACGT[4] = {'A', 'C', 'G', 'T'};

/* length / genome start / strand / emitted length of reads idx[0..n) (idx == NULL: first, first + 1, ...) */
int cb_host_meta(const cb_params *p, const uint32_t *lentab, const uint64_t *idx, uint64_t first, uint64_t n,
                 uint32_t *len, uint64_t *start, uint8_t *strand, uint32_t *out_len) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)n; ++t) {
        const uint64_t i = idx ? idx[t] : first + (uint64_t)t;
        const uint64_t rk = cb_read_key(p, i);
        const uint32_t l = cb_read_len(p, rk, lentab);
        const uint64_t s = cb_read_start(p, rk, l);
        const uint32_t st = cb_read_strand(rk);
        if (len) len[t] = l;
        if (start) start[t] = s;
        if (strand) strand[t] = (uint8_t)st;
        if (out_len) {
            uint32_t o = 0, c0, c1;
            for (uint32_t j = 0; j < l; ++j) o += cb_emit(p, rk, s, l, st, j, &c0, &c1);
            out_len[t] = o ? o : 1;
        }
    }
    return 0;
}

/* ASCII bases of reads idx[0..n) into bases + offsets[t] (offsets from the emitted lengths of cb_host_meta) */
int cb_host_write(const cb_params *p, const uint32_t *lentab, const uint64_t *idx, uint64_t first, uint64_t n,
                  const uint64_t *offsets, char *bases) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)n; ++t) {
        const uint64_t i = idx ? idx[t] : first + (uint64_t)t;
        const uint64_t rk = cb_read_key(p, i);
        const uint32_t l = cb_read_len(p, rk, lentab);
        const uint64_t s = cb_read_start(p, rk, l);
        const uint32_t st = cb_read_strand(rk);
        char *o = bases + offsets[t];
        uint64_t k = 0;
        uint32_t c0, c1;
        for (uint32_t j = 0; j < l; ++j) {
            const uint32_t r = cb_emit(p, rk, s, l, st, j, &c0, &c1);
            if (r >= 1) o[k++] = CB_ACGT[c0];
            if (r == 2) o[k++] = CB_ACGT[c1];
        }
        if (k == 0) {
            uint32_t code = cb_genome_base(p, st ? s + l - 1 : s);
            o[0] = CB_ACGT[st ? 3u - code : code];
        }
    }
    return 0;
}

/* the genome itself, [pos, pos + n) as ASCII (truth for the tests) */
int cb_host_genome(const cb_params *p, uint64_t pos, uint64_t n, char *out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)n; ++t) out[t] = CB_ACGT[cb_genome_base(p, pos + (uint64_t)t)];
    return 0;
}
