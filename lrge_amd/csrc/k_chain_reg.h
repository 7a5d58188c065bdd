// k_chain_reg.h -- K6, register-window form (the hot kernel).
//
// Same arithmetic as chain_group<> in k_chain.h (mg_lchain_dp + mg_chain_backtrack), organised around
// what the data show: the predecessor loop of mg_lchain_dp ends after <= 32 candidates for ~all
// anchors (the max_skip break), so the candidates of anchor i are almost always the 64 anchors just
// before it.  One wavefront owns one (query, target, strand) group and keeps those 64 anchors in
// VGPRs -- lane L holds anchor i-1-L as {x, y, f, p|span} -- shifting them by one lane per step with
// a DPP wave_shr.  A step therefore needs no LDS or HBM read for its candidates:
//   * anchor i itself comes from a 64-anchor staging register filled by one coalesced HBM load every
//     64 steps (v_readlane with a scalar lane index);
//   * comput_sc runs on all lanes; the scalar loop's order-dependent state is resolved with DPP
//     scans exactly as in k_chain.h;
//   * the t[] marks ("candidate j is the predecessor of a candidate already visited") are one
//     ds_permute_b32: every valid candidate pushes a 1 to the lane that holds its predecessor;
//   * (f, p) go to HBM once (8 bytes per anchor) for the backtrack.
// No LDS is allocated, so occupancy is bounded by registers only and every group size runs in the
// same kernel.  When the 64-anchor window is exhausted without a break (or the max_ii shortcut has
// to rescan a longer window) an exact slow path continues through HBM.
#pragma once
#include "k_chain.h"

#define GREC_NONE 0xFFFFFFu   // "no predecessor" in the 24-bit p field

// per-anchor record written by the DP: f (bits 0..31) | p (bits 32..55) | backtrack state (bits 56..63)
__device__ __forceinline__ u64 grec_make(i32 f, i32 p) { return (u64)(u32)f | (u64)((u32)p & GREC_NONE) << 32; }
__device__ __forceinline__ i32 grec_f(u64 r) { return (i32)(u32)r; }
__device__ __forceinline__ i32 grec_p(u64 r) { u32 p = (u32)(r >> 32) & GREC_NONE; return p == GREC_NONE ? -1 : (i32)p; }
__device__ __forceinline__ u32 grec_state(u64 r) { return (u32)(r >> 56); }

// Loads of data this wave stored earlier (grec, tmark).  A group is private to one wavefront, and a
// CU's L1 is coherent with that CU's own stores once they have drained (vmcnt), so ordinary cached
// loads are correct; `volatile` only stops the compiler from reusing a value across our stores.
__device__ __forceinline__ u64 ld_u64_l2(const u64 *p) { return *(const volatile u64 *)p; }
__device__ __forceinline__ u32 ld_u32_l2(const u32 *p) { return *(const volatile u32 *)p; }
__device__ __forceinline__ void drain_stores() { __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00); }  // vmcnt(0)

struct RegChainArgs {
    const u64 *akey, *aval;
    const u32 *gstart;
    u32 n_groups; u64 n_anchors;
    const u32 *list;        // [N_BINS][n_groups] group ids per size bin
    u32 bin_first[N_BINS];  // block index at which each bin starts (largest bin first)
    u32 bin_of[N_BINS];     // which bin that is
    u32 n_blocks;
    u64 *grec;              // [n_anchors]
    u32 *tmark;             // [n_anchors] zero-initialised; slow-path t[] stamps
};

// resolve one 64-candidate chunk of the scalar predecessor loop (used by the out-of-line slow path).
// Inputs are per lane (lane order = visiting order); scalar state is passed by reference.  Returns
// true when the loop broke.
__device__ __forceinline__ bool resolve_chunk(i32 s, bool valid, bool marked, i32 base, i32 max_skip, i32 &max_f,
                                              i32 &max_j, i32 &n_skip, i32 &end_j) {
    i32 exc = wave_shr1_i32(wave_incl_max_i32(s, SC_NONE), SC_NONE);
    exc = exc > max_f ? exc : max_f;
    const bool improve = valid && s > exc;
    const bool bump = valid && !improve && marked;
    const u64 im_all = __ballot(improve);
    const u64 bm_all = __ballot(bump);
    u64 brk;
    i32 ns_after;
    if (im_all == 0) {
        ns_after = n_skip + (i32)__builtin_amdgcn_mbcnt_hi((u32)(bm_all >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm_all, 0)) + (bump ? 1 : 0);
        brk = __ballot(bump && ns_after > max_skip);
    } else {
        i32 a = improve ? -1 : (bump ? 1 : 0);
        i32 b = improve ? 0 : NEG_BIG;
        wave_incl_clampadd(a, b, NEG_BIG);
        ns_after = n_skip + a; ns_after = ns_after > b ? ns_after : b;
        brk = __ballot(bump && ns_after > max_skip);
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
    if (brk) { end_j = base - bl; return true; }
    n_skip = __builtin_amdgcn_readlane(ns_after, 63);
    return false;
}

struct GroupView {   // what the out-of-line paths need to reach a group's data in HBM
    const u64 *gk, *gv; u64 *grec; u32 *tmark; u64 rmask;
};

// Slow path 1: the scalar predecessor loop of anchor i runs past the 64 anchors held in registers
// (no break inside the window and older anchors still in reach).  Continues exactly, through HBM.
// far_push/far_p: lanes whose mark target fell outside the window.  Returns st (window start) or -1
// when the loop broke before the start was determined.
struct SlowTail { i32 st, max_f, max_j, end_j; };   // returned by value: nothing in the hot loop may be address-taken
__device__ __noinline__ SlowTail chain_slow_tail(GroupView V, ChainParams P, i32 i, i32 xi, i32 yi, i32 lower, bool far_push,
                                                 i32 far_p, i32 max_f, i32 max_j, i32 n_skip, i32 first_base) {
    const i32 lane = (i32)lane_id();
    const u32 stamp = (u32)i + 1;
    i32 end_j = lower - 1, st = lower;
    if (far_push) V.tmark[far_p] = stamp;
    drain_stores();
    for (i32 base = first_base; base >= lower; base -= 64) {
        const i32 jj = base - lane;
        const bool inb = jj >= lower;
        i32 xj = 0, yj = 0, sj = 0, fj = 0, pj = -1;
        if (inb) {
            const u64 k = V.gk[jj], v = V.gv[jj], r = ld_u64_l2(V.grec + jj);
            xj = (i32)(k & V.rmask); yj = (i32)(u32)v; sj = (i32)((v >> 32) & 0xff); fj = grec_f(r); pj = grec_p(r);
        }
        const bool reach = inb && xj + P.max_dist_x >= xi;
        const i32 n_reach = (i32)__popcll(__ballot(reach));
        if (n_reach == 0) { st = base + 1; end_j = base; break; }
        const i32 sc2 = comput_sc_dev(xi, yi, xj, yj, sj, P);
        const bool valid2 = reach && sc2 != SC_NONE;
        const i32 s2 = valid2 ? sc2 + fj : SC_NONE;
        if (valid2 && pj >= 0) V.tmark[pj] = stamp;
        drain_stores();
        const bool marked2 = valid2 && ld_u32_l2(V.tmark + jj) == stamp;
        if (resolve_chunk(s2, valid2, marked2, base, P.max_skip, max_f, max_j, n_skip, end_j)) { st = -1; break; }
        if (n_reach < 64) { st = base - n_reach + 1; end_j = st - 1; break; }
    }
    SlowTail r; r.st = st; r.max_f = max_f; r.max_j = max_j; r.end_j = end_j;
    return r;
}

// Slow path 2: max_ii must be re-derived over a window longer than the registers.  `best` holds the
// per-lane candidates from the register window; returns the wave-wide best key (f << 32 | j).
__device__ __noinline__ u64 chain_slow_rescan(GroupView V, ChainParams P, i32 i, i32 xi, i32 lower, i32 st, u64 best,
                                              i32 first_base) {
    const i32 lane = (i32)lane_id();
    if (st < 0) {   // window start not determined yet
        st = lower;
        for (i32 base = first_base; base >= lower; base -= 64) {
            const i32 jj = base - lane;
            const bool reach = jj >= lower && (i32)(V.gk[jj >= 0 ? jj : 0] & V.rmask) + P.max_dist_x >= xi;
            const i32 n_reach = (i32)__popcll(__ballot(reach));
            if (n_reach < 64) { st = base - n_reach + 1; break; }
        }
    }
    drain_stores();
    for (i32 jj = first_base - lane; jj >= st; jj -= 64) {
        const u64 key = (u64)(u32)grec_f(ld_u64_l2(V.grec + jj)) << 32 | (u32)jj;
        best = key > best ? key : best;
    }
    return best;
}

__global__ __launch_bounds__(64) void k_chain_reg(RegChainArgs R, ChainParams P, GroupOut out) {
    if (blockIdx.x >= R.n_blocks) return;
    const i32 lane = (i32)lane_id();
    // block -> (bin, slot): bins are laid out largest first so that long groups start early
    u32 bsel = 0;
#pragma unroll
    for (int b = 1; b < N_BINS; ++b) if (blockIdx.x >= R.bin_first[b]) bsel = b;
    const u32 g = RFL(R.list[(u64)R.bin_of[bsel] * R.n_groups + (blockIdx.x - R.bin_first[bsel])]);
    const u32 s0 = RFL(R.gstart[g]);
    const u32 e0 = (g + 1 < R.n_groups) ? RFL(R.gstart[g + 1]) : (u32)R.n_anchors;
    const i32 n = (i32)(e0 - s0);
    const u64 *gk = R.akey + s0, *gv = R.aval + s0;
    u64 *grec = R.grec + s0;
    u32 *tmark = R.tmark + s0;
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    const u64 k0 = gk[0];
    const u32 rev = RFL((u32)(k0 >> P.kl.sh_rev()) & 1);
    const u32 rid = RFL((u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1));
    const u32 qid = RFL(P.q0 + (u32)(k0 >> P.kl.sh_q()));
    GroupView V; V.gk = gk; V.gv = gv; V.grec = grec; V.tmark = tmark; V.rmask = rmask;

    // ---------------- DP ----------------
    // Written to keep the shared scalar ALU idle: predicates live in the score itself (s == SC_NONE
    // means "not a candidate"), selects replace branches, and (f, p) leave through a staging VGPR.
    const i32 maxdx = P.max_dist_x, bw = P.bw, max_skip = P.max_skip, max_iter = P.max_iter;
    const u32 dqlim = (u32)(P.max_dist_x < P.max_dist_y ? P.max_dist_x : P.max_dist_y);
    const float pen_gap = P.pen_gap, pen_skip = P.pen_skip;
    i32 wx = 0, wy = 0, wf = 0, wp = -1, ws = 0;     // window: lane L <-> anchor i-1-L
    i32 sx = 0, sy = 0, ss = 0;                      // staging in:  lane L <-> anchor (i & ~63) + L
    i32 of = 0, op = -1;                             // staging out: lane L <-> (f, p) of anchor (i & ~63) + L
    i32 mi = -1, mi_x = 0, mi_y = 0, mi_f = 0, mi_sp = 0;
    for (i32 i = 0; i < n; ++i) {
        const i32 il = i & 63;
        if (il == 0) {
            const i32 idx = i + lane;
            if (idx < n) { const u64 k = gk[idx], v = gv[idx]; sx = (i32)(k & rmask); sy = (i32)(u32)v; ss = (i32)((v >> 32) & 0xff); }
        }
        const i32 xi = __builtin_amdgcn_readlane(sx, il), yi = __builtin_amdgcn_readlane(sy, il), spi = __builtin_amdgcn_readlane(ss, il);
        const i32 lim = i - 1 < max_iter - 1 ? i - 1 : max_iter - 1;     // lanes 0..lim hold anchors >= lower
        // reach: x ascending, so the candidates in reach are a prefix of the lanes
        i32 actv = wx + (maxdx - xi);
        actv = lane <= lim ? actv : -1;
        const u64 act_mask = __ballot(actv >= 0);
        const i32 n_act = (i32)__popcll(act_mask);
        // comput_sc, branch-free (same operations in the same order as comput_sc_dev)
        const i32 dq = yi - wy, dr = xi - wx;
        const i32 df = dr - dq;
        const i32 dd = df < 0 ? -df : df;
        const i32 dg = dr < dq ? dr : dq;
        i32 sc = ws < dg ? ws : dg;
        const float lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
        float log_pen = mg_log2_dev((float)(dd + 1));
        log_pen = dd >= 1 ? log_pen : 0.0f;
        const i32 pen = (i32)(lin_pen + .5f * log_pen);
        sc = (dd != 0 || dg > ws) ? sc - pen : sc;
        i32 s = sc + wf;
        s = (u32)(dq - 1) < dqlim ? s : SC_NONE;     // 0 < dq <= min(max_dist_x, max_dist_y)
        s = dr != 0 ? s : SC_NONE;
        s = dd <= bw ? s : SC_NONE;
        s = actv >= 0 ? s : SC_NONE;
        // t[p[j]] = i: every candidate pushes a flag to the lane that holds its predecessor.  Lanes
        // with nothing to push send 0 to lane 0, which is never a real target (p[j] <= i-2).
        const i32 tl = (i - 1) - wp;                 // = lane + (j - p[j]); >= 64 or "p = -1" -> not in the window
        const bool in_reg = s != SC_NONE && (u32)tl < 64u && wp >= 0;
        const i32 got = __builtin_amdgcn_ds_permute(in_reg ? tl << 2 : 0, in_reg ? 1 : 0);
        // running maximum before each lane
        i32 exc = wave_shr1_i32(wave_incl_max_i32(s, SC_NONE), SC_NONE);
        exc = exc > spi ? exc : spi;
        const u64 im_all = __ballot(s > exc);                            // improving candidates
        const i32 bumpv = (s != SC_NONE && s <= exc && got != 0) ? 1 : 0;
        const u64 bm_all = __ballot(bumpv != 0);
        // n_skip after each lane: x -> max(x + d, 0) with d = +1 (bump) / -1 (improve), x0 = 0
        //   = S(L) - min(0, min_{t<=L} S(t)),  S = prefix sums of d  (Lindley recursion)
        i32 ns = (i32)__builtin_amdgcn_mbcnt_hi((u32)(bm_all >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm_all, 0)) + bumpv;
        if (im_all != 0) {
            const i32 impv = s > exc ? 1 : 0;
            const i32 S = ns - ((i32)__builtin_amdgcn_mbcnt_hi((u32)(im_all >> 32), __builtin_amdgcn_mbcnt_lo((u32)im_all, 0)) + impv);
            i32 mn = wave_incl_max_i32(-S, SC_NONE);                     // -min_{t<=L} S(t)
            mn = mn > 0 ? mn : 0;
            ns = S + mn;
        }
        const u64 brk = __ballot(bumpv != 0 && ns > max_skip);
        i32 max_f = spi, max_j = -1, end_j = i - n_act - 1;
        u64 im = im_all;
        if (brk) {
            const i32 bl = (i32)__ffsll((unsigned long long)brk) - 1;
            end_j = i - 1 - bl;
            im &= (2ULL << bl) - 1;                                       // lanes visited before the break
        }
        if (im) {
            const i32 L = 63 - (i32)__clzll((long long)im);
            max_f = __builtin_amdgcn_readlane(s, L);
            max_j = i - 1 - L;
        }
        i32 st = i - n_act;                            // exact unless the window reaches past the registers
        const bool beyond = n_act == 64 && i - 65 >= (i - max_iter > 0 ? i - max_iter : 0);
        if (beyond) {
            const i32 lower = i - max_iter > 0 ? i - max_iter : 0;
            st = -1;
            if (!brk) {
                const bool far_push = s != SC_NONE && wp >= 0 && tl >= 64;
                const SlowTail r = chain_slow_tail(V, P, i, xi, yi, lower, far_push, wp, max_f, max_j,
                                                   __builtin_amdgcn_readlane(ns, 63), i - 65);
                st = r.st; max_f = r.max_f; max_j = r.max_j; end_j = r.end_j;
            }
        }
        // ---- max_ii bookkeeping (the "best f in the window" shortcut) ----
        if (mi < 0 || xi - mi_x > maxdx) {
            u64 best = actv >= 0 ? ((u64)(u32)wf << 32 | (u32)(i - 1 - lane)) : 0;   // f > 0; ties keep the larger j
            if (beyond) best = chain_slow_rescan(V, P, i, xi, i - max_iter > 0 ? i - max_iter : 0, st, best, i - 65);
            best = wave_max_u64(best);
            const u32 bhi = RFL((u32)(best >> 32)), blo = RFL((u32)best);
            if (bhi == 0) mi = -1;
            else {
                mi = (i32)blo; mi_f = (i32)bhi;
                const i32 d = i - 1 - mi;
                if (d < 64) { mi_x = __builtin_amdgcn_readlane(wx, d); mi_y = __builtin_amdgcn_readlane(wy, d); mi_sp = __builtin_amdgcn_readlane(ws, d); }
                else { const u64 k = gk[mi], v = gv[mi]; mi_x = RFL((i32)(k & rmask)); mi_y = RFL((i32)(u32)v); mi_sp = RFL((i32)((v >> 32) & 0xff)); }
            }
        }
        if (mi >= 0 && mi < end_j) {
            const i32 tmp = RFL(comput_sc_dev(xi, yi, mi_x, mi_y, mi_sp, P));
            if (tmp != SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = mi; }
        }
        if (mi < 0 || (xi - mi_x <= maxdx && mi_f < max_f)) { mi = i; mi_x = xi; mi_y = yi; mi_f = max_f; mi_sp = spi; }
        // results: into the out-staging lane, flushed coalesced every 64 anchors
        of = lane == il ? max_f : of;
        op = lane == il ? max_j : op;
        if (il == 63 || i == n - 1) {
            const i32 idx = (i & ~63) + lane;
            if (idx <= i) grec[idx] = grec_make(of, op);
        }
        // shift the window by one lane and insert anchor i at lane 0
        wx = __builtin_amdgcn_update_dpp(xi, wx, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wy = __builtin_amdgcn_update_dpp(yi, wy, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wf = __builtin_amdgcn_update_dpp(max_f, wf, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wp = __builtin_amdgcn_update_dpp(max_j, wp, DPP_WAVE_SHR1, 0xf, 0xf, false);
        ws = __builtin_amdgcn_update_dpp(spi, ws, DPP_WAVE_SHR1, 0xf, 0xf, false);
    }
    drain_stores();

    // ---------------- backtrack (mg_chain_backtrack; commentary in k_chain.h) ----------------
    // state lives in the top byte of grec: 0 free, 1 claimed, 4 "end visited while unclaimed".
    // mg_chain_bk_end's temporary t[]=2 marks are omitted: p[i] < i strictly, so a walk cannot meet its
    // own nodes again and the marks (set, then reset before anyone else reads them) change nothing.
    // The cut (max_i), its score (= max_s) and its anchor count are tracked during that single walk,
    // so when only "is there an accepted chain" is asked, no second walk is needed.
    u8 *gstate = (u8 *)grec;
    u32 flags = 0;
    const i32 qlen = (i32)P.q_len[qid], tlen = (i32)P.t_len[rid];
    const bool need_records = out.chains != nullptr || P.remove_internal != 0;
    for (;;) {
        u64 best = 0;
        for (i32 i = n - 1 - lane; i >= 0; i -= 64) {
            const u64 r = ld_u64_l2(grec + i);
            const i32 fi = grec_f(r);
            if (fi >= P.min_sc && grec_state(r) == 0) { const u64 key = (u64)(u32)fi << 32 | (u32)i; best = key > best ? key : best; }
        }
        best = wave_max_u64(best);
        const i32 zx = (i32)RFL((u32)(best >> 32));
        if (zx == 0) break;
        const i32 top = (i32)RFL((u32)best);
        // walk 1 (mg_chain_bk_end): find where the chain is cut
        i32 i = top, max_i = top, max_s = 0, depth = 0, cnt = 0;
        u64 r = ld_u64_l2(grec + top);
        for (;;) {
            i = RFL(grec_p(r));
            ++depth;
            i32 s;
            if (i < 0) s = zx;
            else { r = ld_u64_l2(grec + i); s = zx - RFL(grec_f(r)); }
            if (s > max_s) { max_s = s; max_i = i; cnt = depth; }
            else if (max_s - s > P.max_drop) break;
            if (i < 0 || (RFL(grec_state(r)) & 3) != 0) break;
        }
        const i32 sc = max_i == top ? 0 : max_s;       // = zx - f[max_i] (or zx at the chain start)
        const bool accepted = sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt;
        if (accepted && !need_records) { flags = 3u; break; }   // count-only: first accepted chain decides
        // walk 2: claim top -> (exclusive) max_i; coordinates and mm_cal_fuzzy_len for accepted chains
        i32 first = top, mlen = 0, blen = 0;
        for (i = top; i != max_i;) {
            if (lane == 0) gstate[(u64)i * 8 + 7] = 1;
            first = i;
            const i32 pi = RFL(grec_p(ld_u64_l2(grec + i)));
            if (accepted && pi != max_i) {
                const u64 ki = gk[i], vi = gv[i], kp = gk[pi], vp = gv[pi];
                const i32 span = RFL((i32)((vi >> 32) & 0xff));
                const i32 tl = RFL((i32)(ki & rmask)) - RFL((i32)(kp & rmask)), ql = RFL((i32)(u32)vi) - RFL((i32)(u32)vp);
                blen += tl > ql ? tl : ql;
                mlen += (tl > span && ql > span) ? span : (tl < ql ? tl : ql);
            }
            i = pi;
        }
        if (cnt == 0 && lane == 0) gstate[(u64)top * 8 + 7] = 4;
        drain_stores();
        if (accepted) {
            const u64 kf = gk[first], vf = gv[first], kt = gk[top], vt = gv[top];
            const i32 fx = RFL((i32)(kf & rmask)), fy = RFL((i32)(u32)vf), q_span = RFL((i32)((vf >> 32) & 0xff));
            const i32 tx = RFL((i32)(kt & rmask)), ty = RFL((i32)(u32)vt);
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
                    c.qs = qs; c.qe = qe; c.rs = rs; c.re = re; c.mlen = mlen; c.blen = blen;
                    {   // entries of minimap2's mini_pos[] spanned by the chain (kept-seed ranks ride in the anchor values)
                        const i32 r0 = (i32)(vf >> AVAL_RANK_SHIFT), r1 = (i32)(vt >> AVAL_RANK_SHIFT);
                        c.n_seeds = (r1 > r0 ? r1 - r0 : r0 - r1) + 1;
                    }
                    out.chains[slot] = c;
                }
            }
            if (!P.want_all && (flags & 2u)) break;
        }
    }
    if (lane == 0) out.flags[g] = flags;
}
