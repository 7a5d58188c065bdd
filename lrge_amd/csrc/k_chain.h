// k_chain.h -- K6 chaining DP + backtrack, K7 overlap counting.
//
// Restates minimap2's mg_lchain_dp / comput_sc / mg_log2 / mg_chain_bk_end / mg_chain_backtrack
// (mm2:lchain.c, mm2:mmpriv.h) and mm_reg_set_coor / mm_cal_fuzzy_len (mm2:hit.c) per
// (query, target, strand) GROUP.  Facts used:
//   * a predecessor always shares (strand, rid) with its successor, and `st` / `max_ii` reset at a
//     group boundary, so the DP over the whole sorted anchor array factorises into independent
//     per-group DPs with group-local indices;
//   * the backtrack visits chain ends by descending f (ties: larger index first under the stable
//     tie policy) and skips ends already claimed, which equals "repeatedly take the best unclaimed
//     end".
// One wavefront owns one group.  The inner predecessor loop is evaluated 64 candidates at a time:
// every lane scores one candidate, then the scalar loop's order-dependent state (running max,
// n_skip with its clamp at 0, the t[] marks, the max_skip break) is resolved with wave scans, so the
// result is bit-identical to the sequential loop.  f32 penalties are computed without contraction
// (the file is compiled with -ffp-contract=off).
#pragma once
#include "internal.h"
#include "k_prims.h"
#include "k_seed.h"

struct ChainParams {
    i32 max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, max_drop;
    float pen_gap, pen_skip;
    KeyLayout kl;
    u32 q0;                     // first query of the batch
    int remove_internal;        // 0 off, 1 forward/AVA predicate (mapping.rs:59-77), 2 inverse (twoset.rs:493-517)
    float max_overhang_ratio;
    int want_all;               // 1: walk every chain (PAF / -F); 0: stop at the first accepted chain
    const u32 *q_len, *t_len;
};

#define SC_NONE INT32_MIN
#define NEG_BIG (-(1 << 29))

__device__ __forceinline__ float mg_log2_dev(float x) {  // mm2:mmpriv.h:mg_log2 (x >= 2)
    u32 z = __float_as_uint(x);
    float log_2 = (float)(i32)(((z >> 23) & 255) - 128);
    z &= ~(255u << 23);
    z += 127u << 23;
    float m = __uint_as_float(z);
    log_2 += (-0.34484843f * m + 2.02466578f) * m - 0.67487759f;
    return log_2;
}

__device__ __forceinline__ i32 comput_sc_dev(i32 xi, i32 yi, i32 xj, i32 yj, i32 spanj, const ChainParams &P) {
    i32 dq = yi - yj;
    if (dq <= 0 || dq > P.max_dist_x) return SC_NONE;
    i32 dr = xi - xj;
    if (dr == 0 || dq > P.max_dist_y) return SC_NONE;
    i32 dd = dr > dq ? dr - dq : dq - dr;
    if (dd > P.bw) return SC_NONE;
    i32 dg = dr < dq ? dr : dq;
    i32 sc = spanj < dg ? spanj : dg;
    if (dd || dg > spanj) {
        float lin_pen = P.pen_gap * (float)dd + P.pen_skip * (float)dg;
        float log_pen = dd >= 1 ? mg_log2_dev((float)(dd + 1)) : 0.0f;
        sc -= (i32)(lin_pen + .5f * log_pen);
    }
    return sc;
}

// per-group working set; P16 = u16 predecessor/mark arrays (LDS variant, n <= 65534)
template <typename IdxT>
struct GroupMem {
    i32 *X, *Y, *F;
    IdxT *P, *T;
    u8 *S;
};

template <typename IdxT> __device__ __forceinline__ i32 ld_idx(IdxT v);
template <> __device__ __forceinline__ i32 ld_idx<u16>(u16 v) { return v == 0xFFFF ? -1 : (i32)v; }
template <> __device__ __forceinline__ i32 ld_idx<i32>(i32 v) { return v; }

struct GroupOut {
    u32 *flags;             // [n_groups] bit0: some chain accepted, bit1: some accepted chain kept after -F
    lrge_hip_chain *chains; // optional record sink
    unsigned long long *n_chains;
    u64 chain_cap;
};

template <typename IdxT, bool LDS>
__device__ void chain_group(const GroupMem<IdxT> M, i32 n, const ChainParams &P, u32 g, u32 qid, u32 rid, u32 rev,
                            const GroupOut &out) {
    const i32 lane = (i32)lane_id();
    i32 st = 0, max_ii = -1;

    // ---------------- DP ----------------
    for (i32 i = 0; i < n; ++i) {
        const i32 xi = M.X[i], yi = M.Y[i];
        while (st < i && xi > M.X[st] + P.max_dist_x) ++st;
        if (i - st > P.max_iter) st = i - P.max_iter;
        i32 max_f = (i32)M.S[i], max_j = -1, n_skip = 0, end_j = st - 1;
        for (i32 base = i - 1; base >= st; base -= 64) {
            const i32 j = base - lane;
            const bool act = j >= st;
            i32 sc = SC_NONE, fj = 0, pj = -1;
            if (act) {
                sc = comput_sc_dev(xi, yi, M.X[j], M.Y[j], (i32)M.S[j], P);
                fj = M.F[j];
                pj = ld_idx<IdxT>(M.P[j]);
            }
            const bool valid = act && sc != SC_NONE;
            const i32 s = valid ? sc + fj : SC_NONE;
            // every candidate that is reached marks its own predecessor (t[p[j]] = i)
            if (valid && pj >= 0) M.T[pj] = (IdxT)i;
            if (!LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            else __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave have landed
            const bool marked = valid && (i32)M.T[j] == i;
            // running maximum before each lane (exclusive prefix max, seeded with max_f)
            i32 exc = wave_shr1_i32(wave_incl_max_i32(s, SC_NONE), SC_NONE);
            exc = exc > max_f ? exc : max_f;
            const bool improve = valid && s > exc;
            const bool bump = valid && !improve && marked;
            // n_skip as a composition of x -> max(x + a, b): improve = (-1, 0), bump = (+1, -inf)
            i32 a = improve ? -1 : (bump ? 1 : 0);
            i32 b = improve ? 0 : NEG_BIG;
            wave_incl_clampadd(a, b, NEG_BIG);
            i32 ns_after = n_skip + a; ns_after = ns_after > b ? ns_after : b;
            const u64 brk = __ballot(bump && ns_after > P.max_skip);
            u64 consider = ~0ULL;
            i32 bl = 64;
            if (brk) { bl = __ffsll((unsigned long long)brk) - 1; consider = (bl == 63) ? ~0ULL : ((1ULL << (bl + 1)) - 1); }
            const u64 im = __ballot(improve) & consider;
            if (im) {
                const i32 L = 63 - __clzll((long long)im);
                max_f = __builtin_amdgcn_readlane(s, L);
                max_j = base - L;
            }
            if (brk) { end_j = base - bl; break; }
            n_skip = __builtin_amdgcn_readlane(ns_after, 63);
        }
        // max_ii bookkeeping (the "best f in the window" shortcut)
        if (max_ii < 0 || xi - M.X[max_ii] > P.max_dist_x) {
            i32 best = SC_NONE, bj = -1;
            for (i32 j = i - 1 - lane; j >= st; j -= 64) { i32 fj = M.F[j]; if (fj > best) { best = fj; bj = j; } }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                i32 ob = __shfl_xor(best, d, 64), oj = __shfl_xor(bj, d, 64);
                if (ob > best || (ob == best && oj > bj)) { best = ob; bj = oj; }
            }
            max_ii = bj;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            i32 tmp = comput_sc_dev(xi, yi, M.X[max_ii], M.Y[max_ii], (i32)M.S[max_ii], P);
            if (tmp != SC_NONE) { i32 cand = tmp + M.F[max_ii]; if (max_f < cand) { max_f = cand; max_j = max_ii; } }
        }
        if (lane == 0) { M.F[i] = max_f; M.P[i] = (IdxT)max_j; }
        if (!LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        else __builtin_amdgcn_s_waitcnt(0xc07f);
        if (max_ii < 0 || (xi - M.X[max_ii] <= P.max_dist_x && M.F[max_ii] < max_f)) max_ii = i;
    }

    // ---------------- backtrack ----------------
    for (i32 i = lane; i < n; i += 64) M.T[i] = 0;
    if (!LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else __builtin_amdgcn_s_waitcnt(0xc07f);
    u32 flags = 0;
    const i32 qlen = (i32)P.q_len[qid], tlen = (i32)P.t_len[rid];
    for (;;) {
        // best unclaimed chain end: max f (>= min_sc), ties -> larger index
        i32 best = SC_NONE, bi = -1;
        for (i32 i = n - 1 - lane; i >= 0; i -= 64) {
            i32 fi = M.F[i];
            if (fi >= P.min_sc && M.T[i] == 0 && fi > best) { best = fi; bi = i; }  // state 0, not visited
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            i32 ob = __shfl_xor(best, d, 64), oi = __shfl_xor(bi, d, 64);
            if (ob > best || (ob == best && oi > bi)) { best = ob; bi = oi; }
        }
        if (bi < 0) break;
        const i32 top = bi, zx = best;
        // mg_chain_bk_end (uniform scalar walk; lane 0 writes the marks)
        i32 i = top, end_i = -1, max_i = top, max_s = 0;
        do {
            if (lane == 0) M.T[i] = (IdxT)((M.T[i] & 4) | 2);
            i = ld_idx<IdxT>(M.P[i]);
            end_i = i;
            i32 s = i < 0 ? zx : zx - M.F[i];
            if (s > max_s) { max_s = s; max_i = i; }
            else if (max_s - s > P.max_drop) break;
            if (!LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            else __builtin_amdgcn_s_waitcnt(0xc07f);
        } while (i >= 0 && (M.T[i] & 3) == 0);
        for (i = top; i >= 0 && i != end_i; i = ld_idx<IdxT>(M.P[i])) if (lane == 0) M.T[i] = (IdxT)(M.T[i] & 4);
        // claim the chain top -> (exclusive) max_i, accumulating mm_cal_fuzzy_len on the way
        i32 cnt = 0, first = top, mlen = 0, blen = 0;
        for (i = top; i != max_i; ) {
            if (lane == 0) M.T[i] = 1;
            ++cnt; first = i;
            const i32 pi = ld_idx<IdxT>(M.P[i]);
            if (pi != max_i) {  // step (pi -> i) lies inside the chain
                const i32 span = (i32)M.S[i];
                const i32 tl = M.X[i] - M.X[pi], ql = M.Y[i] - M.Y[pi];
                blen += tl > ql ? tl : ql;
                mlen += (tl > span && ql > span) ? span : (tl < ql ? tl : ql);
            }
            i = pi;
        }
        // An end whose walk claims nothing keeps state 0 in the scalar code (later walks may pass
        // through it) but is never visited again: bit 2 takes it out of the selection only.
        if (cnt == 0 && lane == 0) M.T[top] = 4;
        if (!LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        else __builtin_amdgcn_s_waitcnt(0xc07f);
        const i32 sc = i < 0 ? zx : zx - M.F[i];
        if (sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt) {
            // mm_reg_set_coor
            const i32 q_span = (i32)M.S[first];
            const i32 rs = M.X[first] + 1 > q_span ? M.X[first] + 1 - q_span : 0;
            const i32 re = M.X[top] + 1;
            i32 qs, qe;
            if (!rev) { qs = M.Y[first] + 1 - q_span; qe = M.Y[top] + 1; }
            else { qs = qlen - (M.Y[top] + 1); qe = qlen - (M.Y[first] + 1 - q_span); }
            mlen += q_span; blen += q_span;
            bool keep = true;
            if (P.remove_internal) {
                i32 overhang = !rev ? min(qs, rs) + min(qlen - qe, tlen - re) : min(qs, tlen - re) + min(qlen - qe, rs);
                i32 maplen = max(qe - qs, re - rs);
                if (P.remove_internal == 1) {
                    float ratio = (float)overhang / (float)maplen;  // IEEE division (no fast-math)
                    if (ratio < P.max_overhang_ratio) keep = false;
                } else {
                    float prod = (float)maplen * P.max_overhang_ratio;
                    i32 lim = prod != prod ? 0 : (prod >= 2147483648.0f ? INT32_MAX : (prod <= -2147483648.0f ? INT32_MIN : (i32)prod));
                    if (overhang > lim) keep = false;
                }
            }
            flags |= 1u | (keep ? 2u : 0u);
            if (out.chains && lane == 0) {
                unsigned long long slot = atomicAdd(out.n_chains, 1ULL);
                if (slot < out.chain_cap) {
                    lrge_hip_chain c;
                    c.query = qid; c.target = rid; c.rev = (i32)rev; c.score = sc; c.cnt = cnt;
                    c.qs = qs; c.qe = qe; c.rs = rs; c.re = re; c.mlen = mlen; c.blen = blen; c.n_seeds = 0;   /* only the hw/reg kernels fill it */
                    out.chains[slot] = c;
                }
            }
            if (!P.want_all && (flags & 2u)) break;
        }
    }
    if (lane == 0) out.flags[g] = flags;
}

// ------------------------------------------------------------------------------------------
// Scalarised LDS variant (the hot kernel).  Same arithmetic as chain_group<>, restructured for the
// CDNA scalar unit: every wave-uniform quantity (i, st, max_f, n_skip, max_ii and its anchor) is
// forced into SGPRs with readfirstlane/readlane so that control flow is s_cbranch on SCC instead of
// exec-mask juggling, each anchor is one 16-byte LDS record {x, y, f, p | span << 16} fetched with a
// single ds_read_b128, and the window start advances with one ballot per anchor.
// ------------------------------------------------------------------------------------------
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

struct __attribute__((aligned(16))) AnchorRec { i32 x, y, f; u32 ps; };  // ps = p (0xFFFF none) | span << 16

__device__ __forceinline__ i32 rec_p(u32 ps) { u32 p = ps & 0xFFFFu; return p == 0xFFFFu ? -1 : (i32)p; }

__global__ __launch_bounds__(64) void k_chain_lds(const u64 *__restrict__ akey, const u64 *__restrict__ aval,
                                                  const u32 *__restrict__ gstart, u32 n_groups, u64 n_anchors,
                                                  const u32 *__restrict__ list, u32 n_list, u32 cap, ChainParams P,
                                                  GroupOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= n_list) return;
    const i32 lane = (i32)lane_id();
    const u32 g = RFL(list[blockIdx.x]);
    const u32 s0 = RFL(gstart[g]);
    const u32 e0 = (g + 1 < n_groups) ? RFL(gstart[g + 1]) : (u32)n_anchors;
    const i32 n = (i32)(e0 - s0);
    AnchorRec *A = (AnchorRec *)smem;
    u16 *T = (u16 *)(A + cap);
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    for (i32 i = lane; i < n; i += 64) {
        u64 k = akey[s0 + i], v = aval[s0 + i];
        AnchorRec r; r.x = (i32)(k & rmask); r.y = (i32)(u32)v; r.f = 0; r.ps = 0xFFFFu | ((u32)(v >> 32) & 0xffu) << 16;
        A[i] = r; T[i] = 0;
    }
    const u64 k0 = akey[s0];
    const u32 rev = RFL((u32)(k0 >> P.kl.sh_rev()) & 1);
    const u32 rid = RFL((u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1));
    const u32 qid = RFL(P.q0 + (u32)(k0 >> P.kl.sh_q()));
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // ---------------- DP ----------------
    i32 st = 0;
    i32 mi = -1, mi_x = 0, mi_y = 0, mi_f = 0, mi_sp = 0;  // max_ii and its anchor, all scalar
    for (i32 i = 0; i < n; ++i) {
        const AnchorRec ai = A[i];
        const i32 xi = RFL(ai.x), yi = RFL(ai.y), spi = RFL((i32)(ai.ps >> 16));
        // advance the window start: X is ascending, so the lanes that fail form a prefix
        for (;;) {
            const i32 jj = st + lane;
            const bool out_of_reach = jj < i && xi > A[jj].x + P.max_dist_x;
            const u64 m = __ballot(out_of_reach);
            st += (i32)__popcll(m);
            if (m != ~0ULL) break;
        }
        if (i - st > P.max_iter) st = i - P.max_iter;
        i32 max_f = spi, max_j = -1, n_skip = 0, end_j = st - 1;
        for (i32 base = i - 1; base >= st; base -= 64) {
            const i32 j = base - lane;
            const bool act = j >= st;
            const AnchorRec aj = A[act ? j : i];
            const i32 pj = rec_p(aj.ps);
            i32 sc = comput_sc_dev(xi, yi, aj.x, aj.y, (i32)(aj.ps >> 16), P);
            const bool valid = act && sc != SC_NONE;
            const i32 s = valid ? sc + aj.f : SC_NONE;
            if (valid && pj >= 0) T[pj] = (u16)i;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const bool marked = valid && (i32)T[act ? j : i] == i;
            i32 exc = wave_shr1_i32(wave_incl_max_i32(s, SC_NONE), SC_NONE);
            exc = exc > max_f ? exc : max_f;
            const bool improve = valid && s > exc;
            const bool bump = valid && !improve && marked;
            const u64 im_all = __ballot(improve);
            const u64 bm_all = __ballot(bump);
            u64 brk;
            i32 ns_after;
            if (im_all == 0) {
                // no candidate improves: n_skip only grows, by the number of bumps at or below the lane
                ns_after = n_skip + (i32)__builtin_amdgcn_mbcnt_hi((u32)(bm_all >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm_all, 0)) + (bump ? 1 : 0);
                brk = __ballot(bump && ns_after > P.max_skip);
            } else {
                i32 a = improve ? -1 : (bump ? 1 : 0);
                i32 b = improve ? 0 : NEG_BIG;
                wave_incl_clampadd(a, b, NEG_BIG);
                ns_after = n_skip + a; ns_after = ns_after > b ? ns_after : b;
                brk = __ballot(bump && ns_after > P.max_skip);
            }
            u64 consider = ~0ULL;
            i32 bl = 64;
            if (brk) { bl = (i32)__ffsll((unsigned long long)brk) - 1; consider = (bl == 63) ? ~0ULL : ((1ULL << (bl + 1)) - 1); }
            const u64 im = im_all & consider;
            if (im) {
                const i32 L = 63 - (i32)__clzll((long long)im);
                max_f = __builtin_amdgcn_readlane(s, L);
                max_j = base - L;
            }
            if (brk) { end_j = base - bl; break; }
            n_skip = __builtin_amdgcn_readlane(ns_after, 63);
        }
        // max_ii bookkeeping
        if (mi < 0 || xi - mi_x > P.max_dist_x) {
            u64 best = 0;  // f > 0 always, so (f << 32 | j) > 0 for any real candidate
            for (i32 j = i - 1 - lane; j >= st; j -= 64) {
                const u64 key = (u64)(u32)A[j].f << 32 | (u32)j;
                best = key > best ? key : best;   // per-lane j descends: equal f keeps the larger j
            }
            best = wave_max_u64(best);
            const u32 bhi = RFL((u32)(best >> 32)), blo = RFL((u32)best);
            if (bhi == 0) mi = -1;
            else {
                mi = (i32)blo;
                const AnchorRec am = A[mi];
                mi_x = RFL(am.x); mi_y = RFL(am.y); mi_f = RFL(am.f); mi_sp = RFL((i32)(am.ps >> 16));
            }
        }
        if (mi >= 0 && mi < end_j) {
            const i32 tmp = RFL(comput_sc_dev(xi, yi, mi_x, mi_y, mi_sp, P));
            if (tmp != SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = mi; }
        }
        if (lane == 0) { A[i].f = max_f; A[i].ps = ((u32)max_j & 0xFFFFu) | (u32)spi << 16; }
        if (mi < 0 || (xi - mi_x <= P.max_dist_x && mi_f < max_f)) { mi = i; mi_x = xi; mi_y = yi; mi_f = max_f; mi_sp = spi; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // ---------------- backtrack (mg_chain_backtrack; see chain_group<> for the commentary) ----------------
    for (i32 i = lane; i < n; i += 64) T[i] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    u32 flags = 0;
    const i32 qlen = (i32)P.q_len[qid], tlen = (i32)P.t_len[rid];
    for (;;) {
        u64 best = 0;
        for (i32 i = n - 1 - lane; i >= 0; i -= 64) {
            const i32 fi = A[i].f;
            if (fi >= P.min_sc && T[i] == 0) { const u64 key = (u64)(u32)fi << 32 | (u32)i; best = key > best ? key : best; }
        }
        best = wave_max_u64(best);
        const i32 zx = (i32)RFL((u32)(best >> 32));
        if (zx == 0) break;
        const i32 top = (i32)RFL((u32)best);
        i32 i = top, end_i = -1, max_i = top, max_s = 0;
        do {
            if (lane == 0) T[i] = (u16)((T[i] & 4) | 2);
            i = RFL(rec_p(A[i].ps));
            end_i = i;
            const i32 s = i < 0 ? zx : zx - RFL(A[i].f);
            if (s > max_s) { max_s = s; max_i = i; }
            else if (max_s - s > P.max_drop) break;
            __builtin_amdgcn_s_waitcnt(0xc07f);
        } while (i >= 0 && (RFL((i32)T[i]) & 3) == 0);
        for (i = top; i >= 0 && i != end_i; i = RFL(rec_p(A[i].ps))) if (lane == 0) T[i] = (u16)(T[i] & 4);
        i32 cnt = 0, first = top, mlen = 0, blen = 0;
        for (i = top; i != max_i;) {
            if (lane == 0) T[i] = 1;
            ++cnt; first = i;
            const AnchorRec ci = A[i];
            const i32 pi = RFL(rec_p(ci.ps));
            if (pi != max_i) {
                const AnchorRec cp = A[pi];
                const i32 span = RFL((i32)(ci.ps >> 16));
                const i32 tl = RFL(ci.x) - RFL(cp.x), ql = RFL(ci.y) - RFL(cp.y);
                blen += tl > ql ? tl : ql;
                mlen += (tl > span && ql > span) ? span : (tl < ql ? tl : ql);
            }
            i = pi;
        }
        if (cnt == 0 && lane == 0) T[top] = 4;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const i32 sc = i < 0 ? zx : zx - RFL(A[i].f);
        if (sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt) {
            const AnchorRec af = A[first], at = A[top];
            const i32 fx = RFL(af.x), fy = RFL(af.y), q_span = RFL((i32)(af.ps >> 16)), tx = RFL(at.x), ty = RFL(at.y);
            const i32 rs = fx + 1 > q_span ? fx + 1 - q_span : 0;
            const i32 re = tx + 1;
            i32 qs, qe;
            if (!rev) { qs = fy + 1 - q_span; qe = ty + 1; }
            else { qs = qlen - (ty + 1); qe = qlen - (fy + 1 - q_span); }
            mlen += q_span; blen += q_span;
            bool keep = true;
            if (P.remove_internal) {
                i32 overhang = !rev ? min(qs, rs) + min(qlen - qe, tlen - re) : min(qs, tlen - re) + min(qlen - qe, rs);
                i32 maplen = max(qe - qs, re - rs);
                if (P.remove_internal == 1) {
                    float ratio = (float)overhang / (float)maplen;
                    if (ratio < P.max_overhang_ratio) keep = false;
                } else {
                    float prod = (float)maplen * P.max_overhang_ratio;
                    i32 lim = prod != prod ? 0 : (prod >= 2147483648.0f ? INT32_MAX : (prod <= -2147483648.0f ? INT32_MIN : (i32)prod));
                    if (overhang > lim) keep = false;
                }
            }
            flags |= 1u | (keep ? 2u : 0u);
            if (out.chains && lane == 0) {
                unsigned long long slot = atomicAdd(out.n_chains, 1ULL);
                if (slot < out.chain_cap) {
                    lrge_hip_chain c;
                    c.query = qid; c.target = rid; c.rev = (i32)rev; c.score = sc; c.cnt = cnt;
                    c.qs = qs; c.qe = qe; c.rs = rs; c.re = re; c.mlen = mlen; c.blen = blen; c.n_seeds = 0;   /* only the hw/reg kernels fill it */
                    out.chains[slot] = c;
                }
            }
            if (!P.want_all && (flags & 2u)) break;
        }
    }
    if (lane == 0) out.flags[g] = flags;
}

// Global-memory variant for groups too large for LDS: scratch arrays are indexed like the anchors.
__global__ __launch_bounds__(64) void k_chain_glb(const u64 *__restrict__ akey, const u64 *__restrict__ aval,
                                                  const u32 *__restrict__ gstart, u32 n_groups, u64 n_anchors,
                                                  const u32 *__restrict__ list, u32 n_list, i32 *gX, i32 *gY, i32 *gF,
                                                  i32 *gP, i32 *gT, u8 *gS, ChainParams P, GroupOut out) {
    if (blockIdx.x >= n_list) return;
    const u32 g = list[blockIdx.x];
    const u32 s0 = gstart[g];
    const u64 e0 = (g + 1 < n_groups) ? gstart[g + 1] : n_anchors;
    const i32 n = (i32)(e0 - s0);
    GroupMem<i32> M;
    M.X = gX + s0; M.Y = gY + s0; M.F = gF + s0; M.P = gP + s0; M.T = gT + s0; M.S = gS + s0;
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    for (i32 i = (i32)lane_id(); i < n; i += 64) {
        u64 k = akey[s0 + i], v = aval[s0 + i];
        M.X[i] = (i32)(k & rmask); M.Y[i] = (i32)(u32)v; M.S[i] = (u8)(v >> 32);
        M.T[i] = 0; M.P[i] = -1; M.F[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const u64 k0 = akey[s0];
    const u32 rev = (u32)(k0 >> P.kl.sh_rev()) & 1;
    const u32 rid = (u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1);
    const u32 qid = P.q0 + (u32)(k0 >> P.kl.sh_q());
    chain_group<i32, false>(M, n, P, g, qid, rid, rev, out);
}

// ------------------------------------------------------------------------------------------
// K7: counting.  Groups are ordered (query, target, strand), so the two strands of one pair are
// adjacent; a pair is counted once (HashSet of target names, twoset.rs:286-302).
// mode 0: two-set forward  -> counts[query]++ per distinct target NAME, has_map[query] = any chain
// mode 1: inverse          -> counts[target]++ per streamed read (twoset.rs:520-523)
// mode 2: all-vs-all       -> counts[query]++ and counts[target]++ per pair, self skipped (ava.rs:277-301)
// ------------------------------------------------------------------------------------------
struct CountParams {
    KeyLayout kl; u32 q0; int mode;
    const u32 *q_rank, *t_rank;   // may be null (then names are all distinct)
    int t_dup;                    // the indexed set holds repeated identifiers (forward mode only)
    const u32 *q_map;             // all-vs-all over a SHARD of the reads: query index -> index of the same read in the
                                  // indexed set (null: the query set is the indexed set itself)
};

__global__ void k_count(const u64 *__restrict__ skey, const u32 *__restrict__ gstart, const u32 *__restrict__ gflags,
                        u32 n_groups, CountParams cp, u32 *__restrict__ counts, u32 *__restrict__ has_map) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    u32 fl = gflags[g];
    if (!fl) return;
    u64 k = skey[gstart[g]];
    u64 pair = k >> cp.kl.sh_rid();  // (qlocal, rid)
    u32 rid = (u32)pair & ((1u << cp.kl.bits_rid) - 1);
    u32 q = cp.q0 + (u32)(k >> cp.kl.sh_q());
    bool prev_same = false; u32 pfl = 0;
    if (g > 0) {
        u64 pk = skey[gstart[g - 1]];
        if ((pk >> cp.kl.sh_rid()) == pair) { prev_same = true; pfl = gflags[g - 1]; }
    }
    if (cp.mode == 0 && has_map && (fl & 1u)) has_map[q] = 1u;  // benign same-value race
    if (!(fl & 2u)) return;
    if (prev_same && (pfl & 2u)) return;  // the other strand of this pair already counted it
    if (cp.mode == 0 && cp.t_dup && cp.t_rank) {
        // HashSet<target_name>: an earlier kept group of this query whose target carries the same
        // identifier already inserted the name (forward mode never rejects duplicate ids)
        const u32 tr = cp.t_rank[rid];
        const u64 ql = k >> cp.kl.sh_q();
        for (u32 gg = g; gg-- > 0;) {
            u64 kk = skey[gstart[gg]];
            if ((kk >> cp.kl.sh_q()) != ql) break;
            u32 r2 = (u32)(kk >> cp.kl.sh_rid()) & ((1u << cp.kl.bits_rid) - 1);
            if (r2 != rid && (gflags[gg] & 2u) && cp.t_rank[r2] == tr) return;
        }
    }
    if (cp.mode == 0) atomicAdd(&counts[q], 1u);
    else if (cp.mode == 1) atomicAdd(&counts[rid], 1u);
    else {
        if (cp.q_rank && cp.t_rank && cp.q_rank[q] == cp.t_rank[rid]) return;  // &rid == tname: self
        atomicAdd(&counts[cp.q_map ? cp.q_map[q] : q], 1u);
        atomicAdd(&counts[rid], 1u);
    }
}

// K8: per_read_estimate (estimate.rs:142-157), f32, explicit rounding per operation
__global__ void k_estimate(const u32 *__restrict__ counts, const u32 *__restrict__ lens, u32 n, float avg_len,
                           float n_target, float two_thr, float *__restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 c = counts[i];
    if (c == 0) { out[i] = __uint_as_float(0x7f800000u); return; }
    float rl = (float)lens[i];
    float ratio = __fdiv_rn(n_target, (float)c);
    float t = __fadd_rn(rl, avg_len);
    t = __fsub_rn(t, two_thr);
    t = __fadd_rn(t, 1.0f);
    t = __fmul_rn(ratio, t);
    out[i] = __fadd_rn(rl, t);
}
