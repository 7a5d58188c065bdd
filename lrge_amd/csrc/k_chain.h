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
// The kernels are k_chain_hw.h (two groups per wavefront, DPP scans) and k_chain_lpg.h (one group per lane); this file
// holds what both share: parameters, comput_sc, the output record, and K7 / K8.  f32 penalties are computed without
// contraction (the file is compiled with -ffp-contract=off).
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

struct GroupOut {
    u32 *flags;             // [n_groups] bit0: some chain accepted, bit1: some accepted chain kept after -F
    lrge_hip_chain *chains; // optional record sink
    unsigned long long *n_chains;
    u64 chain_cap;
    u32 rid_base;           // partitioned index: first read of the part in the whole indexed set (records carry global ids)
};

#define RFL(v) __builtin_amdgcn_readfirstlane(v)

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
    u32 rid_base;                 // all-vs-all against one PART of a partitioned index: first read of the part in the whole
                                  // indexed set (counts are keyed by the whole set)
};

// list / n_list: the groups that were chained (all others carry no flags: one lane per CHAINED group instead of one per
// group -- 0.5 M of 109 M at C4)
__global__ void k_count(const u64 *__restrict__ skey, const u32 *__restrict__ gstart, const u32 *__restrict__ gflags,
                        const u32 *__restrict__ list, u32 n_list, CountParams cp, u32 *__restrict__ counts, u32 *__restrict__ has_map) {
    const u32 li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_list) return;
    const u32 g = list[li];
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
        atomicAdd(&counts[rid + cp.rid_base], 1u);
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
