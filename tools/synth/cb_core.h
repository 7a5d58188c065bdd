/*
 * cb_core.h -- counter-based synthetic long-read generator: the arithmetic shared by the host twin (cb_host.c, gcc) and
 * the device twin (cb_hip.hip, hipcc).  TEST / BENCH INFRASTRUCTURE, not part of the product library.
 *
 * Every quantity is a pure function of (seed, read index, position): integer arithmetic only, so any read can be produced
 * alone, on either side, bit for bit.  The read model is that of lrge_amd/synth.py: uniform random
 * genome, reads placed uniformly on either strand, lengths from the platform's distribution (a 65 536-entry inverse-CDF
 * table computed once on the host and handed to both twins as data), independent per-base substitution / insertion /
 * deletion errors.
 *
 *   genome base p           = 2 bits of sm64(gseed + (p >> 5) * GAMMA) at 2 * (p & 31)          (no genome array anywhere)
 *   read i: key  rk         = sm64(rseed + i * GAMMA)
 *           length          = min(lentab[sm64(rk + 1 * GAMMA) & 0xFFFF], gsize)
 *           start           = mulhi64(sm64(rk + 2 * GAMMA), gsize - length + 1)
 *           strand          = sm64(rk + 3 * GAMMA) & 1
 *           source base j   = strand ? 3 - genome[start + length - 1 - j] : genome[start + j]
 *           error draw j    = e = sm64((rk ^ ESALT) + (j + 1) * GAMMA):  u = low 32 bits
 *                             u < t_sub               substitution by (code + 1 + ((e >> 32) % 3)) & 3
 *                             u < t_sub + t_ins       the base, then an inserted base (e >> 34) & 3
 *                             u < t_sub + t_ins + t_del   deleted
 *           a read whose every base is deleted (never in practice) is its first source base.
 */
#ifndef CB_CORE_H
#define CB_CORE_H
#include <stdint.h>

#ifdef __HIPCC__
#define CB_FN __host__ __device__ static inline
#else
#define CB_FN static inline
#endif

#define CB_GAMMA 0x9E3779B97F4A7C15ull
#define CB_ESALT 0xA5A5F00DC0FFEE11ull

typedef struct {
    uint64_t gsize;        /* genome size in bases */
    uint64_t gseed, rseed; /* genome / read seeds */
    uint32_t t_sub, t_ins, t_del; /* error thresholds on a 32-bit uniform draw: rate * 2^32 */
    uint32_t pad;
} cb_params;

CB_FN uint64_t cb_sm64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
CB_FN uint64_t cb_mulhi64(uint64_t a, uint64_t b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
CB_FN uint64_t cb_genome_word(const cb_params *p, uint64_t w) { return cb_sm64(p->gseed + w * CB_GAMMA); }
CB_FN uint32_t cb_genome_base(const cb_params *p, uint64_t pos) { return (uint32_t)(cb_genome_word(p, pos >> 5) >> (2 * (pos & 31))) & 3u; }
CB_FN uint64_t cb_read_key(const cb_params *p, uint64_t i) { return cb_sm64(p->rseed + i * CB_GAMMA); }
CB_FN uint32_t cb_read_len(const cb_params *p, uint64_t rk, const uint32_t *lentab) {
    uint64_t l = lentab[cb_sm64(rk + 1 * CB_GAMMA) & 0xFFFFu];
    return (uint32_t)(l < p->gsize ? l : p->gsize);
}
CB_FN uint64_t cb_read_start(const cb_params *p, uint64_t rk, uint32_t len) { return cb_mulhi64(cb_sm64(rk + 2 * CB_GAMMA), p->gsize - len + 1); }
CB_FN uint32_t cb_read_strand(uint64_t rk) { return (uint32_t)(cb_sm64(rk + 3 * CB_GAMMA) & 1u); }
CB_FN uint64_t cb_err_draw(uint64_t rk, uint32_t j) { return cb_sm64((rk ^ CB_ESALT) + ((uint64_t)j + 1) * CB_GAMMA); }

/* What source base j of a read becomes: number of output bases (0, 1, 2) and their 2-bit codes (low / high). */
CB_FN uint32_t cb_emit(const cb_params *p, uint64_t rk, uint64_t start, uint32_t len, uint32_t strand, uint32_t j, uint32_t *c0, uint32_t *c1) {
    const uint64_t gp = strand ? start + len - 1 - j : start + j;
    uint32_t code = cb_genome_base(p, gp);
    if (strand) code = 3u - code;
    const uint64_t e = cb_err_draw(rk, j);
    const uint32_t u = (uint32_t)e;
    *c0 = code; *c1 = 0;
    if (u < p->t_sub) { *c0 = (code + 1u + (uint32_t)((e >> 32) % 3u)) & 3u; return 1; }
    if (u - p->t_sub < p->t_ins) { *c1 = (uint32_t)(e >> 34) & 3u; return 2; }
    if (u - p->t_sub - p->t_ins < p->t_del) return 0;
    return 1;
}
#endif
