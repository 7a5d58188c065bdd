// k_sketch_tile.h -- K1 in its position-parallel form: one workgroup per tile of ST_G chunks (2048 bases) of a read, one LANE PER
// STEP of mm_sketch's loop (mm2:sketch.c; aligner.rs:181-185, :231-241 are the callers this replaces).
//
// k_sketch.h's form gives every lane a chunk and lets it replay the scalar state machine; the 64 lanes of a wavefront then sit in
// 64 different states and each step pays every path of the loop (the word fetch, both emission sites, the rescan: ~200 issue slots
// per step at k = 19 + HPC).  Here nothing is sequential:
//   1. the tile's words (+ a halo) go to LDS; with HPC the steps -- homopolymer runs -- are found as bit masks per word, counted,
//      scanned, and written out as a step list (start base, code), which is packed into a 2-bit code stream + an ambiguity bit stream;
//      without HPC the packed image itself is those streams;
//   2. lane q forms the k-mer that ENDS at step q by a funnel shift out of the code stream (forward strand = the pair-reversed window,
//      reverse strand = its complement), the valid-step count l from the ambiguity stream, the span from the step list, hashes the
//      smaller strand and leaves x[q] in LDS;
//   3. lane p decides from x[p-w+1 .. p+w] and l[p+1 .. p+w] whether mm_sketch would ever write step p's minimizer out.  The scalar
//      loop's state before step t is a pure function of the last w infos (its running minimum is the right-most smallest of them),
//      which turns its emission sites into three predicates -- tests/sketch_model.py states them in Python and
//      tests/test_sketch_model.py checks them against the oracle's state machine;
//   4. a wavefront owns a chunk -- the minimizers of the steps that END in it, whenever mm_sketch would write them (k_sketch_direct
//      attributes by the step DURING which they are written; the concatenation over a read's chunks is the same list) --: ballot +
//      prefix give every written minimizer its place in the chunk's slot, in step order.
// Output contract = k_sketch_direct's (per-chunk counts, chunk c's entries at tmp[c * SK_CAP ...), *overflow), so the scan, the
// compaction, the sort's slot-reading first pass and the ranged / gated launches of host_sketch.inl go on unchanged.
// HPC only: a tile whose halo does not hold the w + k steps in front of it / the w + 1 step starts behind it (homopolymer runs of
// dozens of bases around a tile edge) marks its chunks with ST_REDO and k_sketch_redo does them the sequential way.
#pragma once
#include "k_sketch.h"

#define ST_THREADS 256
#define ST_G 16                              // chunks per tile
#define ST_BASES (ST_G * SK_CHUNK)           // 2048

template <int K, int W, bool HPC>
struct StCfg {
    static constexpr int LW = W + K;                         // the largest threshold l is ever compared with
    static constexpr int HB = HPC ? 96 : 32;                 // bases loaded in front of the tile (word multiple)
    static constexpr int HF = 32;                            // ... and behind it
    static constexpr int NWORDS = (HB + ST_BASES + HF) / 32;
    static constexpr int NSTEPS = HB + ST_BASES + HF;        // a step per base at most
    static constexpr bool NARROW = !HPC && 2 * K <= 32;
    static_assert(HPC || NARROW, "the non-HPC form is written for 2k <= 32");
    static_assert(LW <= 32 && W <= 8 && K >= W, "window extraction is one 32-bit funnel");
    using XT = typename std::conditional<NARROW, u32, u64>::type;
    // The non-HPC form reads its code stream straight out of s_w (u64[NWORDS + 2] seen as dwords) and its ambiguity stream out of
    // s_m (u32[NWORDS + 2]), with NO slack: the funnel of the last step (q = NSTEPS - 1) fetches dwords d .. d + 2 of the former and
    // d .. d + 1 of the latter.  Tie the array sizes to those reads, so that another ST_G / SK_CHUNK / HF / K fails to compile
    // instead of reading past the arrays.
    static constexpr int C_LAST = (2 * (NSTEPS - 1 - K + 1 + 32)) / 32 + 2;      // largest dword index the k-mer funnel reads
    static constexpr int N_LAST = (NSTEPS - 1 - LW + 1 + 32) / 32 + 1;           // ... and the valid-step funnel
    static_assert(HPC || C_LAST < 2 * (NWORDS + 2), "non-HPC code stream: s_w is too short for the last step's funnel");
    static_assert(HPC || N_LAST < NWORDS + 2, "non-HPC ambiguity stream: s_m is too short for the last step's funnel");
};

template <typename XT> __device__ __forceinline__ XT st_none() { return (XT)~(XT)0; }

// bits [bit, bit + 32) of a little-endian dword stream (bit may be any value >= 0; reads two dwords)
__device__ __forceinline__ u32 st_bits32(const u32 *s, u32 bit) {
    const u32 d = bit >> 5, sh = bit & 31;
    return __builtin_amdgcn_alignbit(s[d + 1], s[d], sh);
}
// reverse the order of the sixteen 2-bit fields of a dword
__device__ __forceinline__ u32 st_pairrev(u32 v) {
    const u32 r = __builtin_bitreverse32(v);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

template <int K, int W, bool HPC, bool INDEX_KEYS, bool PK>
__global__ __launch_bounds__(ST_THREADS) void k_sketch_tile(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                           const u64 *__restrict__ woff, const u32 *__restrict__ lens, ChunkMap cm,
                                                           u32 n_chunks, u32 *__restrict__ counts, u32 *__restrict__ overflow,
                                                           u64 *__restrict__ tmp_x, u64 *__restrict__ tmp_y, u32 pk_pos1, u32 pk_ybits,
                                                           u32 cap, u32 c_base) {
    using C = StCfg<K, W, HPC>;
    using XT = typename C::XT;
    constexpr int LW = C::LW;
    constexpr u64 kmask = (1ULL << (2 * K)) - 1;
    const XT NONE = st_none<XT>();
    // ---- LDS ----
    __shared__ u64 s_w[C::NWORDS + 2];                       // [0] = the word in front of the loaded range (or padding), [j + 1] = word wl + j
    __shared__ u32 s_m[C::NWORDS + 2];
    __shared__ XT s_x[W + C::NSTEPS + W + 2];                // x of step q at [W + q]; NONE in front and behind
    __shared__ u8 s_l[C::NSTEPS + W + 2];                    // min(l, LW) | strand << 7
    // HPC only (sized 1 otherwise)
    __shared__ u64 s_start[HPC ? C::NWORDS : 1];
    __shared__ u32 s_pre[HPC ? C::NWORDS + 1 : 1];           // steps that start in front of word j of the loaded range
    __shared__ u16 s_pos[HPC ? C::NSTEPS + 2 : 1];           // first base of step q, relative to lo; [NS] = hi - lo
    __shared__ u32 s_code4[HPC ? C::NSTEPS / 4 + 8 : 1];     // code of step q (0..3, 4 = ambiguous), a byte each
    __shared__ u32 s_pk[HPC ? C::NSTEPS / 16 + 8 : 1];       // 2-bit code stream of the steps, 32 steps of padding in front
    __shared__ u32 s_nb[HPC ? C::NSTEPS / 32 + 4 : 1];       // ambiguity bit stream of the steps, 32 steps of padding in front

    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u32 cb = c_base + blockIdx.x * ST_G;
    const u32 ce = cb + ST_G < n_chunks ? cb + ST_G : n_chunks;
    for (u32 ca = cb; ca < ce;) {
        // ---- the segment: chunks [ca, cz) of read r ----
        const u32 r = cm.find(ca);
        const u32 cs0 = cm.chunk_start[r], cs1 = cm.chunk_start[r + 1];
        const u32 cz = cs1 < ce ? cs1 : ce;
        const u32 nch = cz - ca;
        const i32 len = (i32)lens[r];
        const u64 word_base = woff[r];
        const i32 s = (i32)(ca - cs0) * SK_CHUNK;
        const i32 e = s + (i32)nch * SK_CHUNK < len ? s + (i32)nch * SK_CHUNK : len;
        const i32 lo = s - C::HB > 0 ? s - C::HB : 0;
        const i32 hi = e + C::HF < len ? e + C::HF : len;
        const i32 wl = lo >> 5, nw = ((hi + 31) >> 5) - wl;     // words wl .. wl + nw - 1
        __syncthreads();                                        // (the LDS of the segment before)
        // ---- 1. words ----
        for (i32 j = (i32)tid; j < nw + 2; j += ST_THREADS) {
            const i32 wa = wl - 1 + j;
            const bool in = wa >= 0 && wa * 32 < len && j <= nw;
            s_w[j] = in ? pack[word_base + (u64)wa] : 0ULL;
            s_m[j] = in ? nmask[word_base + (u64)wa] : ~0u;
        }
        for (u32 j = tid; j < (u32)W; j += ST_THREADS) s_x[j] = NONE;
        __syncthreads();
        // HPC: a chunk owns the steps that END in it (a run may start in the chunk before).  The step that holds base b (a word boundary):
        auto step_at = [&](i32 b) -> i32 {
            const u32 j = (u32)(b - lo) >> 5;
            return (i32)s_pre[HPC ? j : 0] - (i32)(1u - ((u32)s_start[HPC ? j : 0] & 1u));
        };
        const u32 *cstream, *nstream;                           // step q: code at bits 2(q + 32) of cstream, ambiguity at bit q + 32 of nstream
        i32 NS;                                                 // steps in the loaded range
        bool redo = false;
        if constexpr (HPC) {
            // step starts of every loaded word (k_sketch.h, RunWords::set)
            if ((i32)tid < nw) {
                const u64 w = s_w[tid + 1]; const u32 m = s_m[tid + 1];
                const u64 prev2 = s_w[tid] >> 62; const bool prevN = (s_m[tid] >> 31) != 0;     // (word 0 of the read: the padding says "ambiguous")
                const u64 ns = m ? spread32(m) : 0ULL;
                const u64 x = w ^ (w << 2 | prev2);
                u64 eq = ~(x | x >> 1) & SK_EVEN;
                eq &= ~(ns | ns << 2 | (prevN ? 1ULL : 0ULL));
                u64 starts = ~eq & SK_EVEN;
                const i32 nvalid = len - (wl + (i32)tid) * 32;
                if (nvalid < 32) starts &= nvalid > 0 ? (1ULL << (2 * nvalid)) - 1 : 0ULL;
                s_start[tid] = starts;
                s_pre[tid + 1] = (u32)__popcll(starts);
            }
            if (tid == 0) s_pre[0] = 0;
            __syncthreads();
            u32 pre = 0;
            if ((i32)tid <= nw) for (u32 j = 1; j <= tid; ++j) pre += s_pre[j];
            __syncthreads();
            if ((i32)tid <= nw) s_pre[tid] = pre;
            if (tid < 2) s_pk[tid] = 0u;
            if (tid == 0) s_nb[0] = ~0u;
            __syncthreads();
            NS = (i32)s_pre[nw];
            const i32 q_own0 = step_at(s);
            const i32 q_own1 = e < len ? step_at(e) : NS;
            redo = (lo > 0 && q_own0 < W + K) || (hi < len && NS - q_own1 < W + 2);
            if (!redo) {
                // 2. the step list: two lanes per word
                {
                    const u32 j = tid >> 1, h = tid & 1;
                    if ((i32)j < nw) {
                        const u64 st = s_start[j], w = s_w[j + 1]; const u32 m = s_m[j + 1];
                        u32 bits = h ? (u32)(st >> 32) : (u32)st;
                        u32 q = s_pre[j] + (h ? (u32)__popc((u32)st) : 0u);
                        while (bits) {
                            const u32 b = (u32)__builtin_ctz(bits) + 32 * h;        // even: base b >> 1 of the word
                            bits &= bits - 1;
                            const u32 code = ((m >> (b >> 1)) & 1u) ? 4u : (u32)((w >> b) & 3ULL);
                            s_pos[q] = (u16)(j * 32 + (b >> 1));
                            ((u8 *)s_code4)[q] = (u8)code;
                            ++q;
                        }
                    }
                    if (tid == 0) s_pos[NS] = (u16)(hi - lo);
                    // codes behind the last step: "ambiguous" (never part of a k-mer that is used)
                    for (u32 q2 = (u32)NS + tid; q2 < (((u32)NS + 15u) & ~15u); q2 += ST_THREADS) ((u8 *)s_code4)[q2] = 4;
                }
                __syncthreads();
                // 2b. the streams: a lane per 16 steps
                for (u32 g = tid; g * 16 < (u32)NS; g += ST_THREADS) {
                    u32 pk = 0, nb = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32 c4 = s_code4[g * 4 + i];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const u32 c = (c4 >> (8 * t)) & 0xffu;
                            pk |= (c & 3u) << (2 * (4 * i + t));
                            nb |= (c >> 2) << (4 * i + t);
                        }
                    }
                    s_pk[2 + g] = pk;
                    ((u16 *)s_nb)[2 + g] = (u16)nb;
                }
                {   // the tail of both streams: defined values behind the last group
                    const u32 ng = ((u32)NS + 15u) >> 4;
                    if (tid < 4) s_pk[2 + ng + tid] = 0u;
                    if (tid < 4) ((u16 *)s_nb)[2 + ng + tid] = 0xffffu;
                }
                __syncthreads();
            }
            cstream = s_pk; nstream = s_nb;
        } else {
            NS = hi - lo;
            cstream = (const u32 *)s_w; nstream = s_m;            // (s_w[0] / s_m[0] = the 32 steps in front)
        }
        if (redo) {
            for (u32 i = tid; i < nch; i += ST_THREADS) counts[ca + i] = ST_REDO;
            ca = cz;
            continue;
        }
        // ---- 3. x of every step ----
        for (i32 q = (i32)tid; q < NS + W + 1; q += ST_THREADS) {
            XT x = NONE; u32 lz = 0;
            if (q < NS) {
                // valid steps up to and including q, capped at LW
                const u32 nwin = st_bits32(nstream, (u32)(q - LW + 1 + 32)) & (LW == 32 ? ~0u : (1u << LW) - 1u);
                const u32 l = nwin ? (u32)(LW - 32 + __builtin_clz(nwin)) : (u32)LW;
                lz = l;
                if (l >= (u32)K) {
                    const u32 bit = 2u * (u32)(q - K + 1 + 32);
                    const u32 d = bit >> 5, sh = bit & 31;
                    const u32 d0 = cstream[d], d1 = cstream[d + 1], d2 = cstream[d + 2];
                    const u32 elo = __builtin_amdgcn_alignbit(d1, d0, sh), ehi = __builtin_amdgcn_alignbit(d2, d1, sh);
                    if constexpr (C::NARROW) {
                        constexpr u32 m32 = (u32)kmask;
                        const u32 E = elo & m32;
                        const u32 kr = ~E & m32;
                        const u32 kf = st_pairrev(E) >> (32 - 2 * K);
                        const u32 z = kf < kr ? 0u : 1u;
                        x = (XT)mm_hash32(z ? kr : kf, m32);
                        lz |= z << 7;
                    } else {
                        const u64 E = ((u64)ehi << 32 | elo) & kmask;
                        const u64 kr = ~E & kmask;
                        const u64 kf = ((u64)st_pairrev(elo) << 32 | st_pairrev(ehi & (u32)(kmask >> 32))) >> (64 - 2 * K);
                        const u32 z = kf < kr ? 0u : 1u;
                        lz |= z << 7;
                        u32 span = (u32)K;
                        if constexpr (HPC) span = (u32)s_pos[q + 1] - (u32)s_pos[q - K + 1 > 0 ? q - K + 1 : 0];
                        if (span < 256) x = (XT)(mm_hash64(z ? kr : kf, kmask) << 8 | (u64)span);
                    }
                }
            }
            s_x[W + q] = x;
            s_l[q] = (u8)lz;
        }
        __syncthreads();
        // ---- 4. which steps are written out; 5. their places ----
        const i32 T = hi == len ? NS : 0x7fffffff;               // the read's last step is step T - 1 of the loaded range
        for (u32 ci = wave; ci < nch; ci += ST_THREADS / 64) {
            i32 q0, q1;
            const i32 cs_ = s + (i32)ci * SK_CHUNK, ce_ = cs_ + SK_CHUNK < e ? cs_ + SK_CHUNK : e;
            if constexpr (HPC) { q0 = step_at(cs_); q1 = ce_ < len ? step_at(ce_) : NS; }
            else { q0 = cs_ - lo; q1 = ce_ - lo; }
            const u32 c = ca + ci;
            const u64 slot = (u64)c * SK_CAP;
            u32 base = 0;
            for (i32 qb = q0; qb < q1; qb += 64) {
                const i32 p = qb + (i32)lane;
                bool emit = false, tie = false;
                XT xp = NONE;
                if (p < q1) xp = s_x[W + p];
                if (xp != NONE) {
                    // (A) p is the running minimum at some time and leaves that role written
                    u32 dE = W + 1, lmask = 0;
#pragma unroll
                    for (int d = W; d >= 1; --d) {
                        const XT xr = s_x[W + p + d];
                        if (xr <= xp && p + d < T) dE = (u32)d;
                        if (d < W) tie |= xr == xp;
                    }
#pragma unroll
                    for (int j = 1; j < W; ++j) lmask |= (s_x[W + p - j] >= xp ? 1u : 0u) << (j - 1);
                    i32 tl = (i32)dE - 1 < W - 1 ? (i32)dE - 1 : W - 1;
                    if (T - 1 - p < tl) tl = T - 1 - p;
                    const u32 nl = (u32)(W - 1 - tl);                       // neighbours in front that still share p's last window
                    if ((~lmask & ((1u << nl) - 1u)) == 0u) {
                        if (dE <= (u32)W) emit = (s_l[p + (i32)dE] & 0x7fu) >= (u32)LW;            // a new element that is not larger takes over
                        else if (p + W < T) emit = (s_l[p + W] & 0x7fu) >= (u32)(LW - 1);          // slides out of the window
                        else emit = true;                                                          // the read ends: final flush
                    }
                }
                // (B), (C): equal minima inside one window (the same k-mer twice within w steps) -- rare, and only a lane that has an
                // equal x BEHIND it within the window can be written out by them
                if (__any(tie && !emit)) {
                    if (tie && !emit) {
                        auto X = [&](i32 t) -> XT { return t < T ? s_x[W + t] : NONE; };     // (t >= -W always)
                        auto rms = [&](i32 t) -> i32 {                                        // right-most smallest of steps t-W+1 .. t
                            i32 best = t; XT bx = X(t);
                            for (i32 u = t - 1; u > t - W; --u) { const XT xu = X(u); if (xu < bx) { best = u; bx = xu; } }
                            return best;
                        };
                        for (i32 t = p + 1; t < p + W && t < T && !emit; ++t) {
                            const u32 lt = s_l[t] & 0x7fu;
                            if (lt == (u32)(LW - 1)) {                                        // (B) the first full window's flush
                                const i32 qm = rms(t - 1);
                                if (qm != p && X(qm) == xp) emit = true;
                            }
                            if (!emit && lt >= (u32)(LW - 1) && rms(t - 1) == t - W && X(t) > X(t - W)) {    // (C) the flush behind a rescan
                                const i32 qm = rms(t);
                                if (qm != p && X(qm) == xp) emit = true;
                            }
                        }
                    }
                }
                const u64 bal = __ballot(emit);
                if (emit) {
                    const u32 rank = base + (u32)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                    if (rank < cap) {
                        const u32 z = (u32)(s_l[p] >> 7);
                        u32 i_last;
                        if constexpr (HPC) i_last = (u32)lo + (u32)s_pos[p + 1] - 1u; else i_last = (u32)(lo + p);
                        const u32 y32 = i_last << 1 | z;
                        u64 xfull;
                        if constexpr (C::NARROW) xfull = (u64)xp << 8 | (u64)K; else xfull = (u64)xp;
                        if (PK) tmp_x[slot + rank] = (xfull >> 8) << pk_ybits | (u64)r << pk_pos1 | (u64)y32;
                        else {
                            tmp_x[slot + rank] = INDEX_KEYS ? xfull >> 8 : xfull;
                            tmp_y[slot + rank] = (u64)r << 32 | (u64)y32;
                        }
                    }
                }
                base += (u32)__popcll(bal);
            }
            if (lane == 0) { counts[c] = base; if (base > cap) *overflow = 1u; }
        }
        ca = cz;
    }
}
